/*
 * fiksi_amd — host-side builder: the C surface under the fiksi::System / Element / Constraint
 * mirror (include/fiksi.hpp for C++, the fiksi_amd Python package, a Rust crate per INTEGRATION.md).
 *
 * Mirrors, function for function, the state changes of the reference builder API:
 *   System::new / add_element / add_constraint        fiksi/src/lib.rs:307-445
 *   elements::{Length,Point,Line,Circle}::create      fiksi/src/elements/mod.rs:280,321,365,437
 *   ElementHandle::{fix,unfix,get_value,update_value} fiksi/src/elements/mod.rs:60-98,560-579
 *   constraints::*::create                            fiksi/src/constraints/mod.rs:317-891
 *   ConstraintHandle::{calculate_residual,update_parameter} constraints/mod.rs:88-110,992-1046
 *   Graph::add_constraint / connected components       fiksi/src/graph.rs:178-258
 *   System::solve                                      fiksi/src/lib.rs:464-466
 * Host logic only (no arithmetic beyond bookkeeping); all numerics run on the device through
 * include/fiksi_amd.h.
 */
#ifndef FIKSI_AMD_BUILDER_H
#define FIKSI_AMD_BUILDER_H

#include "fiksi_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fxs_system fxs_system;

/* ElementTag, fiksi/src/elements/mod.rs:487-493 */
typedef enum fxs_element_tag { FXS_LENGTH = 0, FXS_POINT = 1, FXS_LINE = 2, FXS_CIRCLE = 3 } fxs_element_tag;

/* ConstraintTag, fiksi/src/constraints/mod.rs:893-905 */
typedef enum fxs_constraint_tag {
    FXS_POINT_POINT_COINCIDENCE = 0,
    FXS_POINT_POINT_DISTANCE = 1,
    FXS_POINT_POINT_POINT_ANGLE = 2,
    FXS_POINT_LINE_INCIDENCE = 3,
    FXS_POINT_LINE_DISTANCE = 4,
    FXS_POINT_CIRCLE_INCIDENCE = 5,
    FXS_SEGMENT_SEGMENT_LENGTH_EQUALITY = 6,
    FXS_LINE_LINE_ANGLE = 7,
    FXS_LINE_LINE_PARALLELISM = 8,
    FXS_LINE_LINE_PERPENDICULARITY = 9,
    FXS_LINE_CIRCLE_TANGENCY = 10
} fxs_constraint_tag;

int fxs_system_new(fxs_system** out);
void fxs_system_free(fxs_system* s);
uint32_t fxs_system_id(const fxs_system* s);          /* global AtomicU32 counter, lib.rs:308-309 */
uint32_t fxs_num_elements(const fxs_system* s);
uint32_t fxs_num_constraints(const fxs_system* s);
uint32_t fxs_num_variables(const fxs_system* s);
uint32_t fxs_num_expressions(const fxs_system* s);

/* Element creation: returns the element id (>= 0) or a negative fx_status. */
int64_t fxs_length_create(fxs_system* s, double length);
int64_t fxs_point_create(fxs_system* s, double x, double y);
int64_t fxs_line_create(fxs_system* s, uint32_t point1, uint32_t point2);
int64_t fxs_circle_create(fxs_system* s, uint32_t center, uint32_t radius);
int fxs_element_tag_of(const fxs_system* s, uint32_t element);
int fxs_element_fix(fxs_system* s, uint32_t element);
int fxs_element_unfix(fxs_system* s, uint32_t element);
/* Length: 1 value; Point: x,y; Line: p0.x,p0.y,p1.x,p1.y; Circle: cx,cy,r. Returns the count. */
int fxs_element_get_value(const fxs_system* s, uint32_t element, double out[4]);
int fxs_point_update_value(fxs_system* s, uint32_t element, double x, double y);
int fxs_length_update_value(fxs_system* s, uint32_t element, double length);

/* Constraint creation: `elements` are the handles in the order of the Rust `create` signature,
 * `param` the distance / angle (ignored when the constraint has none). Returns the constraint id
 * or a negative fx_status (wrong element type or count == the cases Rust's type system rejects). */
int64_t fxs_constraint_create(fxs_system* s, int constraint_tag, const uint32_t* elements, uint32_t n_elements,
                              double param);
int fxs_constraint_tag_of(const fxs_system* s, uint32_t constraint);
int fxs_constraint_valency(int constraint_tag);       /* constraints/mod.rs:948-990 */
int fxs_constraint_update_parameter(fxs_system* s, uint32_t constraint, double value);

/* Connected components as assemble::solve iterates them (graph.rs:256-258, empties skipped):
 * component id per element (FX_NO_COMPONENT if none) and per constraint. */
int fxs_components(const fxs_system* s, uint32_t* n_components, uint16_t* element_comp, uint16_t* constraint_comp);

/* The geometric graph as the reference keeps it (graph.rs:98-147, lib.rs:123-137): per element its
 * EncodedElement kind and first variable (Length: idx; Point: idx of x; Line: point1_idx; Circle: center_idx),
 * per constraint its valency, first expression and the incident primitive elements handed to
 * Graph::add_constraint (6 slots each, constraint_n_incident used). Any output pointer may be NULL. */
int fxs_export_graph(const fxs_system* s, uint8_t* element_kind, uint32_t* element_idx, uint8_t* constraint_valency,
                     uint32_t* constraint_expr, uint8_t* constraint_n_incident, uint32_t* constraint_incident);

/* The recombination plan Decomposer::RecursiveAssembly solves the System by (analyze/graph/recursive_assembly.rs:164-480),
 * one plan per live component, concatenated. Words: n_steps, then per step |constraints| ids.. |elements| ids..
 * |free_elements| ids.., then the step's three tables (on_frontiers: element -> clusters; owned_elements: cluster ->
 * elements; frontier_elements: cluster -> elements) each as |entries| (key |list| ids..).. in ascending key order.
 * The reference walks randomly seeded hash sets; this plan walks them in ascending id order (fx_recursive.h).
 * budget: subgraphs the search may grow per call (0 = default). flags: bit0 the reference would panic on this
 * System, bit1 budget exhausted (the reference's search would not finish in reasonable time either).
 * Host only. `out` may be NULL to ask for *length. */
int fxs_recursive_plan(const fxs_system* s, uint64_t budget, uint32_t* out, uint32_t capacity, uint32_t* length, uint32_t* flags);

/* Flat view of n Systems as one fx_batch (arrays owned by the returned object). */
typedef struct fxs_flat fxs_flat;
int fxs_flatten(const fxs_system* const* systems, uint32_t n, fxs_flat** out);
const fx_batch* fxs_flat_batch(const fxs_flat* f);
void fxs_flat_free(fxs_flat* f);
/* Copy solved variables of a flat batch back into the Systems it was built from. */
int fxs_flat_scatter(const fxs_flat* f, fxs_system* const* systems, uint32_t n);

/* System::solve for one System / many independent Systems on the device behind `ctx`.
 * opts->decomposer == 2 (RecursiveAssembly): planned here on the host, each step's cluster problem solved on the
 * device (fx_cluster_solve_batch); fx_result counts are summed over the steps, ncomp = steps solved.
 * FX_ERR_UNSUPPORTED where the reference would panic or its plan search would not finish (fxs_recursive_plan flags). */
int fxs_system_solve(fxs_system* s, fx_ctx* ctx, const fx_solving_opts* opts, fx_result* result);
int fxs_systems_solve(fxs_system* const* systems, uint32_t n, fx_ctx* ctx, const fx_solving_opts* opts,
                      fx_result* results);
/* calculate_residual of every constraint (valency > 1: sqrt of the sum of squares,
 * constraints/mod.rs:99-105); expression residuals come from the device. out: n_constraints. */
int fxs_system_constraint_residuals(const fxs_system* s, fx_ctx* ctx, double* out);

/* System::analyze (lib.rs:455-459): ids of the constraints that over-constrain the System, in expression
 * order (a constraint with two dependent expressions appears twice, as in the reference). `ids` has room
 * for n_expressions entries; *n receives the count. */
int fxs_system_analyze(const fxs_system* s, fx_ctx* ctx, uint32_t* ids, uint32_t* n);

#ifdef __cplusplus
}
#endif
#endif
