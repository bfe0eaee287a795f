"""ctypes bindings for the parity oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under peapods_amd/ imports this module.

`OracleSimulation` mirrors the reference's PyO3 class `IsingSimulation`
(src/lib.rs:104-634): same constructor arguments, `sample()` returning the same
dictionary keys/shapes/dtypes for the options on the hot path, `get_spins()`
and `reset()`.  The disorder aggregation restates
spin-sim/src/statistics/results.rs:165-282 and overlap.rs:106-152 (sequential
f64 sums in realisation order, then one division).
"""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SWEEP_MODES = {"metropolis": 0, "gibbs": 1}
PT_SCHEDULES = {"single_random_edge": 0, "full_ladder": 1}
OC_MODES = {"houdayer": 1, "houd2": 1, "jorg": 2, "cmr": 3, "cmr2": 3}
CLUSTER_MODES = {"wolff": 0, "sw": 1}
ACTIONS = {"update": 0, "observe": 1}
OBSERVATION_KINDS = ("fk", "houdayer", "jorg", "cmr_blue")


class OrcConfig(C.Structure):
    _fields_ = [
        ("n_sweeps", C.c_uint64),
        ("warmup_sweeps", C.c_uint64),
        ("sweep_mode", C.c_int32),
        ("pt_interval", C.c_uint64),
        ("pt_schedule", C.c_int32),
        ("overlap_cluster_interval", C.c_uint64),
        ("overlap_cluster_modes", C.c_int32 * 4),
        ("n_overlap_cluster_modes", C.c_int32),
        ("overlap_cluster_mode", C.c_int32),
        ("n_threads", C.c_int32),
        ("cluster_interval", C.c_uint64),
        ("cluster_mode", C.c_int32),
        ("autocorrelation_max_lag", C.c_uint64),
        ("equilibration_diagnostic", C.c_int32),
        ("collect_cluster_stats", C.c_int32),
        ("cluster_action", C.c_int32),
        ("overlap_cluster_action", C.c_int32),
        ("snapshot_interval", C.c_uint64),
    ]


class OrcAttempt(C.Structure):
    _fields_ = [
        ("edge", C.c_uint64),
        ("accepted", C.c_int),
        ("left_system", C.c_uint64),
        ("right_system", C.c_uint64),
    ]


class OrcRng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4)]


def build(native=False):
    """Compile the oracle with the committed Makefile.  Returns the .so path."""
    out = os.path.join(_HERE, "liboracle_native.so" if native else "liboracle.so")
    args = ["make", "-s", "-C", _HERE, f"OUT={out}"]
    if native:
        args.append("ARCHFLAGS=-march=native")
    subprocess.run(args, check=True)
    return out


def load(path=None):
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    if path is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
    lib = C.CDLL(path)
    u64, i64, u32, i32, f32, vp = C.c_uint64, C.c_int64, C.c_uint32, C.c_int32, C.c_float, C.c_void_p
    P = C.POINTER
    sig = {
        "orc_splitmix64": (u64, [u64]),
        "orc_child_seed": (u64, [u64, u64, u64]),
        "orc_realization_seed": (u64, [u64, u64]),
        "orc_rng_seed_from_u64": (None, [P(OrcRng), u64]),
        "orc_rng_next_u64": (u64, [P(OrcRng)]),
        "orc_rng_next_u32": (u32, [P(OrcRng)]),
        "orc_rng_gen_f32": (f32, [P(OrcRng)]),
        "orc_rng_gen_range_usize": (u64, [P(OrcRng), u64, u64]),
        "orc_rng_shuffle_usize": (None, [P(OrcRng), vp, C.c_size_t]),
        "orc_pin_sample_single_u32": (C.c_size_t, [u32, u32, vp, C.c_size_t, P(u32)]),
        "orc_pin_f32_of_word": (f32, [u32]),
        "orc_pin_sample_single_inclusive_u8": (C.c_size_t, [C.c_uint8, C.c_uint8, vp, C.c_size_t, P(C.c_uint8)]),
        "orc_lattice_new": (vp, [vp, C.c_int]),
        "orc_lattice_with_offsets": (vp, [vp, C.c_int, vp, C.c_size_t]),
        "orc_lattice_free": (None, [vp]),
        "orc_lattice_n_spins": (u64, [vp]),
        "orc_lattice_n_neighbors": (u64, [vp]),
        "orc_lattice_stride": (u64, [vp, C.c_int]),
        "orc_lattice_neighbor_fwd": (u64, [vp, u64, u64]),
        "orc_lattice_neighbor_bwd": (u64, [vp, u64, u64]),
        "orc_lookup_new": (vp, [vp, C.c_size_t, vp, C.c_size_t, C.c_size_t]),
        "orc_lookup_free": (None, [vp]),
        "orc_lookup_accepted_count": (u32, [f32, i32]),
        "orc_lookup_legacy_accepts": (C.c_int, [f32, i32, u32]),
        "orc_lookup_get": (u32, [vp, C.c_size_t, i32]),
        "orc_metropolis_sweep": (None, [vp, vp, vp, vp, vp, C.c_size_t, vp, vp]),
        "orc_gibbs_sweep": (None, [vp, vp, vp, vp, vp, C.c_size_t, vp]),
        "orc_compute_energies": (None, [vp, vp, vp, C.c_size_t, vp, vp, vp]),
        "orc_parallel_tempering": (C.c_size_t, [vp, vp, vp, C.c_size_t, C.c_size_t, P(OrcRng), vp]),
        "orc_parallel_tempering_full_ladder": (
            C.c_size_t,
            [vp, vp, vp, C.c_size_t, C.c_size_t, P(OrcRng), C.c_size_t, vp],
        ),
        "orc_pt_replay": (None, [C.c_size_t, C.c_size_t, vp, vp, C.c_size_t, vp, vp, vp]),
        "orc_uf_from_bonds": (None, [vp, vp, vp, C.c_size_t, vp, vp]),
        "orc_uf_histogram": (None, [vp, C.c_size_t, vp]),
        "orc_top4_sizes": (None, [vp, C.c_size_t, vp]),
        "orc_graph_observation": (None, [vp, vp, vp, C.c_size_t, vp]),
        "orc_sim_get_cluster_stats": (C.c_int, [vp, C.c_size_t, vp, vp, vp]),
        "orc_sim_get_observation": (C.c_int, [vp, C.c_size_t, C.c_int, vp, vp, vp, vp, vp, vp]),
        "orc_sim_n_snapshots": (C.c_size_t, [vp]),
        "orc_sim_get_snapshot": (C.c_int, [vp, C.c_size_t, vp, vp, vp, vp, vp, vp]),
        "orc_sim_create": (
            vp,
            [vp, C.c_int, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, u64],
        ),
        "orc_sim_create_shard": (
            vp,
            [vp, C.c_int, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, u64, u64],
        ),
        "orc_sim_destroy": (None, [vp]),
        "orc_sim_reset": (None, [vp, u64]),
        "orc_sim_sample": (C.c_int, [vp, P(OrcConfig)]),
        "orc_sim_last_error": (C.c_char_p, [vp]),
        "orc_sim_n_spins": (u64, [vp]),
        "orc_sim_n_neighbors": (u64, [vp]),
        "orc_sim_get_spins": (None, [vp, C.c_size_t, vp]),
        "orc_sim_get_system_ids": (None, [vp, C.c_size_t, vp]),
        "orc_sim_get_energies": (None, [vp, C.c_size_t, vp]),
        "orc_sim_get_rng_state": (None, [vp, C.c_size_t, C.c_size_t, vp]),
        "orc_sim_get_pt_counters": (None, [vp, C.c_size_t, vp, vp, vp]),
        "orc_sim_get_result_moments": (None, [vp, C.c_size_t, vp]),
        "orc_sim_has_overlap": (C.c_int, [vp]),
        "orc_sim_get_tau": (C.c_int, [vp, C.c_size_t, vp, vp]),
        "orc_sim_n_equil_checkpoints": (C.c_size_t, [vp]),
        "orc_sim_get_equil": (None, [vp, C.c_size_t, vp, vp, vp]),
        "orc_sim_get_result_overlap": (None, [vp, C.c_size_t, vp]),
        "orc_sim_get_result_histograms": (None, [vp, C.c_size_t, vp, vp, vp]),
    }
    for name, (restype, argtypes) in sig.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if path.endswith("liboracle.so"):
        _LIB = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Lattice:
    """Handle on the oracle's neighbour tables (ss/geometry/lattice.rs)."""

    def __init__(self, shape, offsets=None, lib=None):
        self.lib = lib or load()
        shape_arr = np.asarray(shape, dtype=np.uint64)
        self.shape = tuple(int(x) for x in shape)
        if offsets is None:
            self.handle = self.lib.orc_lattice_new(_ptr(shape_arr), len(shape))
        else:
            off = np.ascontiguousarray(offsets, dtype=np.int64)
            self.handle = self.lib.orc_lattice_with_offsets(
                _ptr(shape_arr), len(shape), _ptr(off), off.shape[0]
            )
        self.n_spins = int(self.lib.orc_lattice_n_spins(self.handle))
        self.n_neighbors = int(self.lib.orc_lattice_n_neighbors(self.handle))

    def strides(self):
        return [int(self.lib.orc_lattice_stride(self.handle, d)) for d in range(len(self.shape))]

    def neighbor_fwd(self, i, d):
        return int(self.lib.orc_lattice_neighbor_fwd(self.handle, i, d))

    def neighbor_bwd(self, i, d):
        return int(self.lib.orc_lattice_neighbor_bwd(self.handle, i, d))

    def __del__(self):
        if getattr(self, "handle", None):
            self.lib.orc_lattice_free(self.handle)
            self.handle = None


def make_rngs(seeds, lib=None):
    lib = lib or load()
    arr = (OrcRng * len(seeds))()
    for i, s in enumerate(seeds):
        lib.orc_rng_seed_from_u64(C.byref(arr[i]), int(s))
    return arr


def aggregate_results(per_real, n_temps, has_overlap, with_samples):
    """SweepResult::aggregate_impl (statistics/results.rs:84-285) + OverlapStats::aggregate_impl
    (statistics/overlap.rs:67-155): sequential sums in realisation order, then /= n."""
    n = float(len(per_real))
    out = {}
    moment_keys = ["mags", "mags2", "mags4", "energies", "energies2"]
    for k, key in enumerate(moment_keys):
        acc = np.zeros(n_temps, dtype=np.float64)
        for res in per_real:
            acc = acc + res["moments"][k]
        out[key] = acc / n if len(per_real) > 1 else acc
    if has_overlap:
        ov_keys = ["overlap", "overlap2", "overlap4", "link_overlap", "link_overlap2", "link_overlap4"]
        for k, key in enumerate(ov_keys):
            acc = np.zeros(n_temps, dtype=np.float64)
            for res in per_real:
                acc = acc + res["overlap"][k]
            out[key] = acc / n if len(per_real) > 1 else acc
        hist = np.zeros_like(per_real[0]["histogram"])
        ql = np.zeros_like(per_real[0]["ql_at_q_sum"])
        ql2 = np.zeros_like(per_real[0]["ql2_at_q_sum"])
        for res in per_real:
            hist = hist + res["histogram"]
            ql = ql + res["ql_at_q_sum"]
            ql2 = ql2 + res["ql2_at_q_sum"]
        out["overlap_histogram"] = [hist[t].copy() for t in range(n_temps)]
        out["ql_at_q_sum"] = ql
        out["ql2_at_q_sum"] = ql2
        if with_samples:
            out["per_sample_overlap_histogram"] = np.stack([r["histogram"] for r in per_real])
            out["per_sample_ql_at_q_sum"] = np.stack([r["ql_at_q_sum"] for r in per_real])
            out["per_sample_ql2_at_q_sum"] = np.stack([r["ql2_at_q_sum"] for r in per_real])
    for key in ("mags2_tau", "overlap2_tau"):  # results.rs:213-228,267-274; src/lib.rs:545-557
        if key in per_real[0]:
            acc = np.zeros(n_temps, dtype=np.float64)
            for res in per_real:
                acc = acc + res[key]
            out[key] = acc / n if len(per_real) > 1 else acc
    if "equil_sweeps" in per_real[0]:  # results.rs:229-244,275-282; src/lib.rs:559-574
        out["equil_sweeps"] = per_real[0]["equil_sweeps"]
        for key in ("equil_energy_avg", "equil_link_overlap_avg"):
            acc = np.zeros_like(per_real[0][key])
            for res in per_real:
                acc = acc + res[key]
            out[key] = acc / n if len(per_real) > 1 else acc
    # ClusterStats (results.rs:181-214,261-267; src/lib.rs:492-543)
    if "fk_csd" in per_real[0]:
        total = np.zeros_like(per_real[0]["fk_csd"])
        for res in per_real:
            total = total + res["fk_csd"]
        if total.any():
            out["fk_csd"] = [total[t].copy() for t in range(n_temps)]
    if "overlap_csd" in per_real[0]:
        total = np.zeros_like(per_real[0]["overlap_csd"])
        tops = np.zeros_like(per_real[0]["top_sizes"])
        for res in per_real:
            total = total + res["overlap_csd"]
            tops = tops + res["top_sizes"]
        if len(per_real) > 1:
            tops = tops / n
        if total.any():
            out["overlap_csd"] = [[total[m, t].copy() for t in range(n_temps)] for m in range(total.shape[0])]
        out["top_cluster_sizes"] = [tops[m].copy() for m in range(tops.shape[0])]
    # per-disorder graph observations: a kind is reported when every realisation has it (src/lib.rs:414-456)
    observations = {}
    for kind in OBSERVATION_KINDS:
        if all(kind in res.get("observations", {}) for res in per_real):
            fields = per_real[0]["observations"][kind].keys()
            observations[kind] = {f: np.stack([res["observations"][kind][f] for res in per_real])
                                  for f in fields}
    if observations:
        out.setdefault("per_disorder", {})["cluster_observations"] = observations
    return out


class OracleSimulation:
    """Mirror of the reference's `IsingSimulation` pyclass (src/lib.rs:104-634)."""

    def __init__(self, lattice_shape, couplings, temperatures, n_replicas=None,
                 neighbor_offsets=None, seed=None, lib=None, first_realization=0):
        self.lib = lib or load()
        self.shape = [int(x) for x in lattice_shape]
        n_dims = len(self.shape)
        n_neighbors = len(neighbor_offsets) if neighbor_offsets is not None else n_dims
        couplings = np.ascontiguousarray(couplings, dtype=np.float32)
        temps = np.ascontiguousarray(temperatures, dtype=np.float32)
        expected_single = tuple(self.shape) + (n_neighbors,)
        if couplings.shape == expected_single:
            n_real = 1
        elif couplings.ndim == len(expected_single) + 1 and couplings.shape[1:] == expected_single:
            n_real = couplings.shape[0]
        else:  # src/lib.rs:146-149
            raise ValueError(
                f"couplings shape {list(couplings.shape)} does not match lattice {list(expected_single)}"
            )
        self.n_replicas = 1 if n_replicas is None else int(n_replicas)
        self.n_temps = len(temps)
        self.n_realizations = n_real
        self.constructor_seed = 42 if seed is None else int(seed)
        shape_arr = np.asarray(self.shape, dtype=np.uint64)
        if neighbor_offsets is not None:
            off = np.ascontiguousarray(neighbor_offsets, dtype=np.int64)
            off_ptr, n_off = _ptr(off), off.shape[0]
        else:
            off_ptr, n_off = None, 0
        self.handle = self.lib.orc_sim_create_shard(
            _ptr(shape_arr), n_dims, off_ptr, n_off, _ptr(couplings), n_real, _ptr(temps),
            self.n_temps, self.n_replicas, self.constructor_seed, int(first_realization),
        )
        self.n_spins = int(self.lib.orc_sim_n_spins(self.handle))
        self.n_systems = self.n_replicas * self.n_temps

    def __del__(self):
        if getattr(self, "handle", None):
            self.lib.orc_sim_destroy(self.handle)
            self.handle = None

    # -- state ------------------------------------------------------------
    def get_spins(self, realization=0):
        out = np.empty(self.n_systems * self.n_spins, dtype=np.int8)
        self.lib.orc_sim_get_spins(self.handle, realization, _ptr(out))
        return out

    def get_system_ids(self, realization=0):
        out = np.empty(self.n_systems, dtype=np.uint64)
        self.lib.orc_sim_get_system_ids(self.handle, realization, _ptr(out))
        return out

    def get_energies(self, realization=0):
        out = np.empty(self.n_systems, dtype=np.float32)
        self.lib.orc_sim_get_energies(self.handle, realization, _ptr(out))
        return out

    def get_rng_state(self, realization, system):
        out = np.empty(4, dtype=np.uint64)
        self.lib.orc_sim_get_rng_state(self.handle, realization, system, _ptr(out))
        return out

    def get_pt_counters(self, realization=0):
        n_edges = max(self.n_temps - 1, 0)
        att = np.zeros(n_edges, dtype=np.uint64)
        acc = np.zeros(n_edges, dtype=np.uint64)
        trips = np.zeros(self.n_systems, dtype=np.uint64)
        self.lib.orc_sim_get_pt_counters(self.handle, realization, _ptr(att), _ptr(acc), _ptr(trips))
        return att, acc, trips.reshape(self.n_replicas, self.n_temps)

    def reset(self, seed=None):
        self.lib.orc_sim_reset(self.handle, self.constructor_seed if seed is None else int(seed))

    # -- sampling ---------------------------------------------------------
    def make_config(self, n_sweeps, sweep_mode, pt_interval=None, pt_schedule=None,
                    overlap_cluster_update_interval=None, overlap_cluster_build_mode=None,
                    overlap_cluster_mode=None, warmup_ratio=None, n_threads=1,
                    cluster_update_interval=None, cluster_mode=None, equilibration_diagnostic=False,
                    autocorrelation_max_lag=None, collect_cluster_stats=False, cluster_action=None,
                    overlap_cluster_action=None, snapshot_interval=None):
        warmup = 0.25 if warmup_ratio is None else warmup_ratio
        cfg = OrcConfig()
        cfg.n_sweeps = int(n_sweeps)
        # Rust f64::round = half away from zero (src/lib.rs:220)
        cfg.warmup_sweeps = int(np.floor(n_sweeps * warmup + 0.5))
        if sweep_mode not in SWEEP_MODES:
            raise ValueError(f"unknown sweep_mode '{sweep_mode}', expected 'metropolis' or 'gibbs'")
        cfg.sweep_mode = SWEEP_MODES[sweep_mode]
        if pt_interval == 0:
            raise ValueError("pt_interval must be >= 1")
        cfg.pt_interval = 0 if pt_interval is None else int(pt_interval)
        sched = pt_schedule or "single_random_edge"
        if sched not in PT_SCHEDULES:
            raise ValueError(
                f"unknown pt_schedule '{sched}', expected 'single_random_edge' or 'full_ladder'"
            )
        cfg.pt_schedule = PT_SCHEDULES[sched]
        if overlap_cluster_update_interval is not None:
            if overlap_cluster_update_interval < 1:
                raise ValueError("overlap_cluster interval must be >= 1")
            cfg.overlap_cluster_interval = int(overlap_cluster_update_interval)
            modes = [m.strip() for m in (overlap_cluster_build_mode or "houdayer").split("+")]
            for k, m in enumerate(modes):
                if m in OC_MODES:
                    cfg.overlap_cluster_modes[k] = OC_MODES[m]
                elif m.startswith("houd") and m[4:].isdigit() and int(m[4:]) % 2 == 0:
                    group = int(m[4:])  # config.rs:124-141; simulation/mod.rs:207-214
                    if group > self.n_replicas:
                        raise ValueError(f"overlap cluster requires n_replicas >= max group_size "
                                         f"({self.n_replicas} < {group})")
                    cfg.overlap_cluster_modes[k] = OC_MODES["houdayer"] | (group << 8)
                else:
                    raise ValueError(f"oracle does not restate overlap mode '{m}'")
            cfg.n_overlap_cluster_modes = len(modes)
            cfg.overlap_cluster_mode = CLUSTER_MODES[overlap_cluster_mode or "wolff"]
        cfg.n_threads = int(n_threads)
        if cluster_update_interval is not None:
            if cluster_update_interval < 1:
                raise ValueError("cluster_update interval must be >= 1")
            cfg.cluster_interval = int(cluster_update_interval)
            cfg.cluster_mode = CLUSTER_MODES[cluster_mode or "sw"]  # src/lib.rs:233
        cfg.equilibration_diagnostic = 1 if equilibration_diagnostic else 0
        cfg.autocorrelation_max_lag = 0 if autocorrelation_max_lag is None else int(autocorrelation_max_lag)
        cfg.collect_cluster_stats = 1 if collect_cluster_stats else 0
        cfg.cluster_action = ACTIONS[cluster_action or "update"]
        cfg.overlap_cluster_action = ACTIONS[overlap_cluster_action or "update"]
        if snapshot_interval is not None and overlap_cluster_update_interval is not None:
            if snapshot_interval < 1:  # config.rs:213-219
                raise ValueError("snapshot_interval must be a positive multiple of overlap_cluster interval")
            cfg.snapshot_interval = int(snapshot_interval)
        return cfg

    def run(self, cfg):
        rc = self.lib.orc_sim_sample(self.handle, C.byref(cfg))
        if rc != 0:
            raise ValueError(self.lib.orc_sim_last_error(self.handle).decode())

    def per_realization_results(self):
        n_t, n_bins = self.n_temps, self.n_spins + 1
        has_ov = bool(self.lib.orc_sim_has_overlap(self.handle))
        per_real = []
        for r in range(self.n_realizations):
            res = {"moments": np.empty((5, n_t), dtype=np.float64)}
            self.lib.orc_sim_get_result_moments(self.handle, r, _ptr(res["moments"]))
            if has_ov:
                res["overlap"] = np.empty((6, n_t), dtype=np.float64)
                self.lib.orc_sim_get_result_overlap(self.handle, r, _ptr(res["overlap"]))
                res["histogram"] = np.empty((n_t, n_bins), dtype=np.uint64)
                res["ql_at_q_sum"] = np.empty((n_t, n_bins), dtype=np.float64)
                res["ql2_at_q_sum"] = np.empty((n_t, n_bins), dtype=np.float64)
                self.lib.orc_sim_get_result_histograms(
                    self.handle, r, _ptr(res["histogram"]), _ptr(res["ql_at_q_sum"]),
                    _ptr(res["ql2_at_q_sum"]),
                )
            m2_tau = np.zeros(n_t, dtype=np.float64)
            q2_tau = np.zeros(n_t, dtype=np.float64)
            got = self.lib.orc_sim_get_tau(self.handle, r, _ptr(m2_tau), _ptr(q2_tau))
            if got >= 1:
                res["mags2_tau"] = m2_tau
            if got >= 2:
                res["overlap2_tau"] = q2_tau
            n_ck = int(self.lib.orc_sim_n_equil_checkpoints(self.handle))
            if n_ck:
                res["equil_sweeps"] = np.empty(n_ck, dtype=np.uint64)
                res["equil_energy_avg"] = np.empty((n_ck, n_t), dtype=np.float64)
                res["equil_link_overlap_avg"] = np.empty((n_ck, n_t), dtype=np.float64)
                self.lib.orc_sim_get_equil(self.handle, r, _ptr(res["equil_sweeps"]),
                                           _ptr(res["equil_energy_avg"]),
                                           _ptr(res["equil_link_overlap_avg"]))
            self._cluster_results(r, res)
            per_real.append(res)
        return per_real, has_ov

    def _cluster_results(self, r, res):
        n_t, n_bins = self.n_temps, self.n_spins + 1
        n_modes = max(getattr(self, "_n_modes", 0), 1)
        fk = np.zeros((n_t, n_bins), dtype=np.uint64)
        ov = np.zeros((n_modes, n_t, n_bins), dtype=np.uint64)
        tops = np.zeros((n_modes, n_t, 4), dtype=np.float64)
        got = self.lib.orc_sim_get_cluster_stats(self.handle, r, _ptr(fk), _ptr(ov), _ptr(tops))
        if got & 1:
            res["fk_csd"] = fk
        if got & 2:
            res["overlap_csd"] = ov
            res["top_sizes"] = tops
        res["observations"] = {}
        for kind, name in enumerate(OBSERVATION_KINDS):
            count = np.zeros(n_t, dtype=np.uint64)
            sizes = np.zeros((n_t, n_bins), dtype=np.uint64)
            top4 = np.zeros((n_t, 4), dtype=np.float64)
            density = np.zeros(n_t, dtype=np.float64)
            large = np.zeros(n_t, dtype=np.float64)
            winding = np.zeros((n_t, 4), dtype=np.float64)
            got = self.lib.orc_sim_get_observation(self.handle, r, kind, _ptr(count), _ptr(sizes), _ptr(top4),
                                                   _ptr(density), _ptr(large), _ptr(winding))
            if not got:
                continue
            obs = {"observation_count": count, "cluster_size_counts": sizes,
                   "top_four_component_fractions": top4, "active_bond_density": density,
                   "large_component_count": large}
            if got & 2:  # src/lib.rs:84-99
                for k, key in enumerate(("winding_x", "winding_y", "winding_either", "winding_both")):
                    obs[key] = winding[:, k].copy()
            res["observations"][name] = obs

    def snapshots(self):
        """`cluster_snapshots` of the last sample() (src/lib.rs:576-614)."""
        out = []
        n_t, n = self.n_temps, self.n_spins
        for k in range(int(self.lib.orc_sim_n_snapshots(self.handle))):
            sweep_id, mode_idx = C.c_uint64(), C.c_int32()
            ids = np.zeros((n_t, n), dtype=np.uint32)
            blue = np.zeros((n_t, n), dtype=np.uint32)
            spins = np.zeros((n_t, 2, n), dtype=np.int8)
            sids = np.zeros((n_t, 2), dtype=np.uint64)
            has_blue = self.lib.orc_sim_get_snapshot(self.handle, k, C.byref(sweep_id), C.byref(mode_idx),
                                                     _ptr(ids), _ptr(blue), _ptr(spins), _ptr(sids))
            snap = {"sweep_id": int(sweep_id.value), "mode_idx": int(mode_idx.value), "cluster_ids": ids,
                    "spins": spins, "system_ids": sids}
            if has_blue:
                snap["blue_ids"] = blue
            out.append(snap)
        return out

    def result_blocks(self):
        """Per-realisation f64 blocks in the layout of pa_sim_get_result_blocks."""
        per_real, has_ov = self.per_realization_results()
        rows = []
        for res in per_real:
            parts = [res["moments"].reshape(-1)]
            if has_ov:
                parts.append(res["overlap"].reshape(-1))
            rows.append(np.concatenate(parts))
        return np.stack(rows)

    def sample(self, n_sweeps, sweep_mode, pt_interval=None, pt_schedule=None,
               overlap_cluster_update_interval=None, overlap_cluster_build_mode=None,
               overlap_cluster_mode=None, warmup_ratio=None, n_threads=1,
               cluster_update_interval=None, cluster_mode=None, equilibration_diagnostic=False,
               autocorrelation_max_lag=None, collect_cluster_stats=False, cluster_action=None,
               overlap_cluster_action=None, snapshot_interval=None):
        cfg = self.make_config(n_sweeps, sweep_mode, pt_interval, pt_schedule,
                               overlap_cluster_update_interval, overlap_cluster_build_mode,
                               overlap_cluster_mode, warmup_ratio, n_threads,
                               cluster_update_interval, cluster_mode, equilibration_diagnostic,
                               autocorrelation_max_lag, collect_cluster_stats, cluster_action,
                               overlap_cluster_action, snapshot_interval)
        self._n_modes = int(cfg.n_overlap_cluster_modes)
        self.run(cfg)
        per_real, has_ov = self.per_realization_results()
        out = aggregate_results(per_real, self.n_temps, has_ov, self.n_realizations > 1)
        if pt_interval is not None:  # src/lib.rs:458-487
            att, acc, trips = zip(*(self.get_pt_counters(r) for r in range(self.n_realizations)))
            out.setdefault("per_disorder", {})["parallel_tempering"] = {
                "edge_attempts": np.stack(att),
                "edge_acceptances": np.stack(acc),
                "round_trips": np.stack(trips),
            }
        snaps = self.snapshots()
        if snaps:
            out["cluster_snapshots"] = snaps
        return out
