#!/usr/bin/env python3
"""Extract the known-answer DATA (inputs + expected outputs) that the reference's own unit tests
hold for the hot path, and store it as JSON fixtures next to this script.

Run in the build container only (needs /root/reference); the GPU box and the test-suite read the
committed JSON files, never the reference tree. Only numeric literals are extracted - no source
text is copied.

  rand.rs:49-63                      -> lcg_known_sequence.json
  qr.rs:468-652                      -> qr_big_underdetermined_damped.json
  cholesky.rs:602-736                -> symbolic_davis_fig1.json
"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def numbers(text):
    out = []
    for x in re.findall(r"-?(?:0x[0-9A-Fa-f]+|\d+\.\d*(?:[eE][-+]?\d+)?|\d+)", text):
        if x.lower().startswith("0x"):
            out.append(int(x, 16))
        elif "." in x or "e" in x.lower():
            out.append(float(x))
        else:
            out.append(int(x))
    return out


def between(src, start_pat, end_pat, pos=0):
    s = re.compile(start_pat).search(src, pos)
    e = re.compile(end_pat).search(src, s.end())
    return src[s.end():e.start()], e.end()


def main():
    # --- LCG -----------------------------------------------------------------------------
    src = open(f"{REF}/fiksi/src/rand.rs").read()
    body, _ = between(src, r"let sequence = \[", r"\];")
    seq = numbers(body)
    json.dump({"source": "fiksi/src/rand.rs:49-63", "seed": seq[0], "sequence": seq[1:]},
              open(f"{HERE}/lcg_known_sequence.json", "w"), indent=1)

    # --- QR big_underdetermined_damped ---------------------------------------------------
    src = open(f"{REF}/solvi/src/decomposition/sparse/qr.rs").read()
    pos = src.index("fn big_underdetermined_damped")
    a_rows, pos = between(src, r"row_indices: vec!\[", r"\],", pos)
    a_cols, pos = between(src, r"column_pointers: vec!\[", r"\],", pos)
    a_vals, pos = between(src, r"values: vec!\[", r"\],", pos)
    b, pos = between(src, r"let b = \[", r"\];", pos)
    r_rows, pos = between(src, r"row_indices: vec!\[", r"\],", pos)
    r_cols, pos = between(src, r"column_pointers: vec!\[", r"\]", pos)
    r_vals, pos = between(src, r"let expected_r_values: &\[f64\] = &\[", r"\];", pos)
    x, pos = between(src, r"let x_expected = \[", r"\];", pos)
    json.dump({
        "source": "solvi/src/decomposition/sparse/qr.rs:468-652",
        "nrows": 21, "ncols": 12,
        "a_row_indices": numbers(a_rows), "a_column_pointers": numbers(a_cols), "a_values": numbers(a_vals),
        "b": numbers(b),
        "r_row_indices": numbers(r_rows), "r_column_pointers": numbers(r_cols),
        "r_abs_values_tol": 1e-8, "r_values": numbers(r_vals),
        "x_expected": numbers(x), "x_tol": 1e-8,
    }, open(f"{HERE}/qr_big_underdetermined_damped.json", "w"), indent=1)

    # --- symbolic: Davis 2011 Fig. 1 -----------------------------------------------------
    src = open(f"{REF}/solvi/src/decomposition/sparse/cholesky.rs").read()
    pos = src.index("fn known_matrix")
    cols_txt, pos = between(src, r"let row_indices: \[&'static \[usize\]; 12\] = \[", r"\];", pos)
    cols = [numbers(c) for c in re.findall(r"&\[([^\]]*)\]", cols_txt)]
    par, pos = between(src, r"parents\.as_slice\(\),\s*&\[", r"\],", pos)
    par = [(-1 if "usize::MAX" in tok else int(tok)) for tok in re.findall(r"usize::MAX|\d+", par)]
    rc, pos = between(src, r"l_counts\.row_counts, &\[", r"\]", pos)
    cc, pos = between(src, r"l_counts\.col_counts, &\[", r"\]", pos)
    lri, pos = between(src, r"&l_structure\.row_indices,\s*&\[", r"\]\s*\);", pos)
    json.dump({
        "source": "solvi/src/decomposition/sparse/cholesky.rs:602-736",
        "nrows": 23, "ncols": 12, "columns": cols, "parents": par,
        "row_counts": numbers(rc), "col_counts": numbers(cc), "l_row_indices": numbers(lri),
    }, open(f"{HERE}/symbolic_davis_fig1.json", "w"), indent=1)
    print("wrote fixtures")


if __name__ == "__main__":
    main()
