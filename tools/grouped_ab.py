"""Diagnostic: the grouped kernel (four Systems per wavefront) against the one-System-per-wavefront kernel
on the same batches — bitwise comparison of results and solved variables, and timing. Each kernel runs in
its own child process (the routing switch FIKSI_AMD_GROUPED is read once per process)."""
import os, subprocess, sys, json, tempfile
import numpy as np

CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, '.')
import fiksi_amd
from fiksi_amd import workloads, abi
sys.path.insert(0, 'tests')
import helpers
ctx = fiksi_amd.Context(0)
out = {}
def run(name, b, **kw):
    db = ctx.upload(b)
    opts = abi.solving_opts(**kw)
    db.system_solve(opts); ctx.synchronize()
    ctx.timer_begin()
    for _ in range(5): db.system_solve(opts)
    ms = ctx.timer_end() / 5
    v = db.get_vars(); r = db.get_results()
    np.save(sys.argv[1] + '_' + name + '_v.npy', v)
    np.save(sys.argv[1] + '_' + name + '_r.npy', r)
    out[name] = ms
    db.free()
n = int(sys.argv[2])
run('ring16', workloads.ring16(n))
run('ring16_f32', workloads.ring16(n), f32=True)
run('hinged11', workloads.hinged_triangles(n, 11))
run('cfg5', workloads.ring16(n, inconsistent=True), f32=True)
run('gauge', workloads.ring16(n, fix_gauge=True))
run('gauge_nopert', workloads.ring16(n, fix_gauge=True), perturb=False)
run('random', workloads.concat([helpers.random_sketch(s).flatten() for s in range(600)]))
run('mixed', workloads.concat([helpers.mixed_sketch(s, fix_some=bool(s & 1)).flatten() for s in range(300)]))
run('mixed_f32', workloads.concat([helpers.mixed_sketch(s, fix_some=bool(s & 1)).flatten() for s in range(300)]), f32=True)
print(json.dumps(out))
'''

def main():
    n = sys.argv[1] if len(sys.argv) > 1 else '20000'
    tmp = tempfile.mkdtemp()
    res = {}
    for tag, env in (('grouped', '1'), ('single', '0')):  # 1 = forced also for small batches
        e = dict(os.environ, FIKSI_AMD_GROUPED=env)
        p = subprocess.run([sys.executable, '-c', CHILD, os.path.join(tmp, tag), n], env=e, capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            print(p.stdout[-2000:], p.stderr[-4000:])
            sys.exit(1)
        res[tag] = json.loads(p.stdout.strip().splitlines()[-1])
    for name in res['grouped']:
        va = np.load(os.path.join(tmp, 'grouped_%s_v.npy' % name)); vb = np.load(os.path.join(tmp, 'single_%s_v.npy' % name))
        ra = np.load(os.path.join(tmp, 'grouped_%s_r.npy' % name)); rb = np.load(os.path.join(tmp, 'single_%s_r.npy' % name))
        same_v = np.array_equal(va.view(np.uint64), vb.view(np.uint64))
        nd = int((va.view(np.uint64) != vb.view(np.uint64)).sum())
        same_r = all(np.array_equal(ra[f], rb[f]) or (ra[f].dtype.kind == 'f' and np.array_equal(ra[f].view(np.uint64), rb[f].view(np.uint64))) for f in ra.dtype.names)
        diff_f = [f for f in ra.dtype.names if not np.array_equal(ra[f], rb[f], equal_nan=True)]
        print(f"{name:12s} grouped {res['grouped'][name]:8.3f} ms  single {res['single'][name]:8.3f} ms  x{res['single'][name]/res['grouped'][name]:.2f}  vars bit-identical: {same_v} ({nd} differ, max |d| {np.nanmax(np.abs(va-vb)):.2e})  results differ in: {diff_f}")

if __name__ == '__main__':
    main()
