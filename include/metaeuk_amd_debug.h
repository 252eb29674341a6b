/* include/metaeuk_amd_debug.h -- entry points of libmetaeuk_amd.so that are NOT part of the drop-in boundary (include/metaeuk_amd.h):
 * the seeded generator of the synthetic workloads (tests, bench.py, tools/) and the hook the experiment kernels of tools/micro/ use.
 * Nothing of the reference binds these; a maintainer wiring the library into src/metaeuk.cpp needs metaeuk_amd.h alone.
 * tests/test_abi_exports.py holds the two headers to the library's export table in BOTH directions: every declared symbol is exported,
 * every exported mk_* symbol is declared in one of them. */
#ifndef METAEUK_AMD_DEBUG_H
#define METAEUK_AMD_DEBUG_H
#include "metaeuk_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- test and bench support (host code): the seeded generator of the synthetic workload of SURVEY.md 8(d) at database sizes the Python
 * generator cannot reach.  Families of ten proteins (founder of 150 .. 600 residues from the Robinson background, member j = the founder
 * with every residue redrawn with probability 0.05 (1 + j)); residues = NULL: only offsets[n + 1] and *total are filled. */
int mk_synth_targets(uint64_t n_targets, uint64_t seed, uint8_t *residues, uint64_t cap, uint64_t *offsets, uint64_t *total);
/* query fragments cut out of the targets ("planted homologs"): fragment k = min_len .. max_len residues of a random target, every residue
 * redrawn with probability mutation_rate; every random_every-th fragment is pure background (0: none).  source[k] = the target it came
 * from (0xFFFFFFFF: background).  residues = NULL: only offsets, source and *total. */
int mk_synth_fragments(uint64_t n_fragments, uint64_t seed, const uint8_t *target_residues, const uint64_t *target_offsets, uint64_t n_targets,
                       double mutation_rate, uint32_t min_len, uint32_t max_len, uint64_t random_every, uint8_t *residues, uint64_t cap,
                       uint64_t *offsets, uint32_t *source, uint64_t *total);
/* residue codes -> an MMseqs2 sequence DB in memory: data[total + 2 n] = "SEQ\n\0" entries, rows of its .index (key = position) */
int mk_synth_seqdb(const uint8_t *residues, const uint64_t *offsets, uint64_t n, char *data, uint32_t *keys, uint64_t *data_offsets, uint32_t *lengths);

/* ... and straight to disk: <base>, <base>.index, <base>.dbtype (amino acids; key = position), piece by piece -- a 2.2e10-residue database is 23 GB of
 * data and 6e7 index rows.  with_lines != 0: <base>.txt too, one sequence per line (what oracle/ref_harness reads). */
int mk_synth_write_seqdb(const char *base, const uint8_t *residues, const uint64_t *offsets, uint64_t n, int with_lines);

/* ---- experiment hook (tools/micro/mk_experiments.hip is its only user): the device view of a (database, batch) pair -- a
 * mk::PrefilterDeviceView (metaeuk_amd/csrc/mk_prefilter.hpp: pointers into HBM, valid while both handles live) copied into view_out, whose
 * size must be given as view_bytes -- and the batch's host offsets.  The layout of that struct is NOT stable across builds. */
int mk_debug_prefilter_view(mk_targetdb *db, mk_queries *q, void *view_out, size_t view_bytes, const uint64_t **q_offsets_host);

#ifdef __cplusplus
}
#endif
#endif
