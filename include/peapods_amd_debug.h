/* Diagnostic entry points of libpeapods_amd.so: in-kernel time stamps of single launches, used by tools/stamps*.py to
 * take a kernel apart (profiles/r0N_stamps_*.txt).  Not part of the drop-in boundary -- nothing in the reference
 * corresponds to them (its Cargo feature `profile` only un-inlines the hot functions for a sampling profiler:
 * spin-sim/Cargo.toml:12-13) -- and nothing in peapods_amd/ calls them.  They advance the ensemble by one sweep like
 * pa_sim_metropolis_sweep does. */
#ifndef PEAPODS_AMD_DEBUG_H
#define PEAPODS_AMD_DEBUG_H

#include "peapods_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* shader-clock stamps of workgroup 0 of one sweep launch (k_sweep / k_sweep_hc): start, loaded, swept, stored, energy, end */
int pa_debug_sweep_stamps(pa_sim *sim, unsigned long long out[6]);
/* s_memrealtime (100 MHz) stamps of EVERY workgroup of one k_sweep_hc launch: out[n_blocks][8]; returns the number of
 * workgroups, 0 when the handle does not use that kernel */
int pa_debug_sweep_hc_stamps(pa_sim *sim, unsigned long long *out, uint64_t capacity_blocks);
/* ... of every wave of one k_sweep_reg launch: out[n_blocks][8] = start, spins loaded, planes landed, levels 0-15 / 16-31 /
 * 32-47 done, stored, energy done; 0 when the handle does not use that kernel */
int pa_debug_sweep_reg_stamps(pa_sim *sim, unsigned long long *out, uint64_t capacity_blocks);
/* s_memrealtime at every level of every workgroup of one k_sweep_slab launch after `settle` ordinary sweeps:
 * out[workgroups][n_levels + 4] = {first level, last level, stamp per level ...}; returns the number of workgroups */
int pa_debug_slab_stamps(pa_sim *sim, unsigned long long *out, uint64_t capacity_words, uint32_t settle);
/* phase stamps of the union-find cluster kernel (k_cluster_sw), every workgroup = one pair / system of one move.
 * enable != 0: the handle's sampling calls stamp from now on (use a handle with one stream group); enable == 0: copy the
 * rows of the last launch, out[workgroups][16] (s_memrealtime, 100 MHz; the slots are listed in tools/stamps_cluster.py),
 * and stop.  Returns the number of workgroups. */
int pa_debug_cluster_stamps(pa_sim *sim, int enable, unsigned long long *out, uint64_t capacity_blocks);
/* shader-clock stamps of the phases of one measurement launch (k_measure) */
int pa_debug_measure_stamps(pa_sim *sim, unsigned long long out[7]);

#ifdef __cplusplus
}
#endif
#endif
