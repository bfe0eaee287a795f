"""`IsingSimulation`: the MI355X counterpart of the reference's PyO3 class
(reference src/lib.rs:104-634).  Same constructor, `sample`, `get_spins` and
`reset`; the ensemble lives on the GPU behind the C ABI of include/peapods_amd.h.
"""

import ctypes as C
import sys
import threading

import numpy as np

from . import _native as nat

_SWEEP_MODES = {"metropolis": 0, "gibbs": 1}
_PT_SCHEDULES = {"single_random_edge": 0, "full_ladder": 1}
_CLUSTER_MODES = {"wolff": 0, "sw": 1}
_CLUSTER_ACTIONS = {"update": 0, "observe": 1}
_AUTOCORR_BACKENDS = {"ring": 0, "fft": 1}
_OC_MODES = {"houdayer": 1, "houd2": 1, "jorg": 2, "cmr": 3, "cmr2": 3}


def _raise(code, text):
    if code == nat.PA_ERR_INTERRUPTED:
        raise KeyboardInterrupt(text)
    if code == nat.PA_ERR_UNSUPPORTED:
        raise NotImplementedError(text)
    if code == nat.PA_ERR_DEVICE:
        raise RuntimeError(text)
    raise ValueError(text)


def _parse_overlap_modes(text):
    """spin-sim/src/config.rs:108-178 (parse_overlap_modes / OverlapClusterBuildMode::try_from)."""
    modes = []
    for part in text.split("+"):
        name = part.strip()
        if name in _OC_MODES:
            modes.append(_OC_MODES[name])
        elif name.startswith("houd"):
            try:
                n = int(name[4:])
            except ValueError:
                raise ValueError(
                    f"invalid Houdayer group size in '{name}', expected 'houdN' with even integer N >= 2"
                ) from None
            if n < 2 or n % 2:
                raise ValueError(f"Houdayer group size must be even and >= 2, got {n}")
            if n > 32:
                raise NotImplementedError(f"houd{n}: groups above 32 replicas are not built on the HIP path")
            if n > 2:  # the reference says so on stderr too (config.rs:135-140); measured: see DESIGN.md
                print(f"WARNING: houd{n} (group_size > 2) is experimental and very likely "
                      "does not satisfy detailed balance", file=sys.stderr)
            modes.append(_OC_MODES["houdayer"] | (n << 8))
        else:
            raise ValueError(
                f"unknown overlap_cluster_build_mode '{name}', expected 'houdayer', 'houdN', "
                "'jorg', or 'cmr'"
            )
    return modes


class IsingSimulation:
    def __init__(self, lattice_shape, couplings, temperatures, n_replicas=None,
                 neighbor_offsets=None, seed=None, device=0, first_realization=0, stream_groups=0):
        self._lib = nat.lib()
        self._handle = None
        shape = [int(x) for x in lattice_shape]
        couplings = np.ascontiguousarray(couplings, dtype=np.float32)
        temps = np.ascontiguousarray(temperatures, dtype=np.float32)
        if temps.ndim != 1:
            raise ValueError("temperatures must be one-dimensional")
        n_neighbors = len(neighbor_offsets) if neighbor_offsets is not None else len(shape)
        expected_single = tuple(shape) + (n_neighbors,)
        if couplings.shape == expected_single:
            n_real = 1
        elif couplings.ndim == len(expected_single) + 1 and couplings.shape[1:] == expected_single:
            n_real = couplings.shape[0]
        else:  # src/lib.rs:146-149
            raise ValueError(
                f"couplings shape {list(couplings.shape)} does not match lattice "
                f"{list(expected_single)}"
            )
        self.lattice_shape = shape
        self.n_replicas = 1 if n_replicas is None else int(n_replicas)
        self.n_temps = int(temps.shape[0])
        self.n_realizations = int(n_real)
        self.constructor_seed = 42 if seed is None else int(seed)  # src/lib.rs:155

        shape_arr = np.asarray(shape, dtype=np.uint64)
        desc = nat.PaSimDesc()
        desc.n_dims = len(shape)
        desc.shape = shape_arr.ctypes.data_as(C.POINTER(C.c_uint64))
        if neighbor_offsets is not None:
            offsets = np.ascontiguousarray(neighbor_offsets, dtype=np.int64)
            if offsets.ndim != 2 or offsets.shape[1] != len(shape):
                raise ValueError("every neighbor offset must have one entry per lattice dimension")
            desc.n_neighbors = offsets.shape[0]
            desc.neighbor_offsets = offsets.ctypes.data_as(C.POINTER(C.c_int64))
        else:
            desc.n_neighbors = 0
            desc.neighbor_offsets = None
        desc.couplings = couplings.ctypes.data_as(C.POINTER(C.c_float))
        desc.n_realizations = n_real
        desc.temperatures = temps.ctypes.data_as(C.POINTER(C.c_float))
        desc.n_temps = self.n_temps
        desc.n_replicas = self.n_replicas
        desc.seed = self.constructor_seed
        desc.device = int(device)
        desc.first_realization = int(first_realization)
        desc.stream_groups = int(stream_groups)  # 0: the default; 1: one stream (what profile_step needs)
        handle = C.c_void_p()
        err = C.create_string_buffer(512)
        code = self._lib.pa_sim_create(C.byref(desc), C.byref(handle), err, len(err))
        if code != nat.PA_OK:
            _raise(code, err.value.decode())
        self._handle = handle
        self.n_spins = int(self._lib.pa_sim_n_spins(handle))
        self.n_neighbors = int(self._lib.pa_sim_n_neighbors(handle))
        self.n_systems = int(self._lib.pa_sim_n_systems(handle))

    def __del__(self):
        if getattr(self, "_handle", None):
            self._lib.pa_sim_destroy(self._handle)
            self._handle = None

    def _check(self, code):
        if code != nat.PA_OK:
            _raise(code, self._lib.pa_sim_last_error(self._handle).decode())

    # ---- configuration ----------------------------------------------------
    def make_config(self, n_sweeps, sweep_mode, cluster_update_interval=None, cluster_mode=None,
                    cluster_action=None, pt_interval=None, pt_schedule=None,
                    overlap_cluster_update_interval=None, overlap_cluster_build_mode=None,
                    overlap_cluster_mode=None, overlap_cluster_action=None, warmup_ratio=None,
                    collect_cluster_stats=None, autocorrelation_max_lag=None,
                    autocorrelation_backend=None, sequential=None, equilibration_diagnostic=None,
                    snapshot_interval=None, measure_overlap=True):
        """Validates the keyword arguments of `sample` (src/lib.rs:219-284, config.rs:180-247)
        without touching any state and returns the C configuration."""
        warmup = 0.25 if warmup_ratio is None else float(warmup_ratio)
        cfg = nat.PaSampleConfig()
        cfg.n_sweeps = int(n_sweeps)
        cfg.warmup_sweeps = int(np.floor(n_sweeps * warmup + 0.5))  # f64::round, src/lib.rs:220
        if sweep_mode not in _SWEEP_MODES:
            raise ValueError(f"unknown sweep_mode '{sweep_mode}', expected 'metropolis' or 'gibbs'")
        cfg.sweep_mode = _SWEEP_MODES[sweep_mode]
        schedule = "single_random_edge" if pt_schedule is None else pt_schedule
        if schedule not in _PT_SCHEDULES:
            raise ValueError(
                f"unknown pt_schedule '{schedule}', expected 'single_random_edge' or 'full_ladder'"
            )
        cfg.pt_schedule = _PT_SCHEDULES[schedule]
        backend = "ring" if autocorrelation_backend is None else autocorrelation_backend
        if backend not in _AUTOCORR_BACKENDS:
            raise ValueError(
                f"unknown autocorrelation_backend '{backend}', expected 'ring' or 'fft'"
            )
        if backend == "fft" and autocorrelation_max_lag is None:
            raise ValueError("autocorrelation_backend='fft' requires autocorrelation_max_lag")
        if pt_interval is not None and pt_interval < 1:
            raise ValueError("pt_interval must be >= 1")
        cfg.pt_interval = 0 if pt_interval is None else int(pt_interval)
        if cluster_update_interval is not None:
            mode = "sw" if cluster_mode is None else cluster_mode
            if mode not in _CLUSTER_MODES:
                raise ValueError(f"unknown cluster_mode '{mode}', expected 'wolff' or 'sw'")
            action = "update" if cluster_action is None else cluster_action
            if action not in _CLUSTER_ACTIONS:
                raise ValueError(f"unknown cluster action '{action}', expected 'update' or 'observe'")
            if cluster_update_interval < 1:
                raise ValueError("cluster_update interval must be >= 1")
            if action == "observe" and mode == "wolff":
                raise ValueError("cluster_action='observe' requires cluster_mode='sw'")
            cfg.cluster_interval = int(cluster_update_interval)
            cfg.cluster_mode = _CLUSTER_MODES[mode]
            cfg.cluster_action = _CLUSTER_ACTIONS[action]
        if overlap_cluster_update_interval is not None:
            modes = _parse_overlap_modes(
                "houdayer" if overlap_cluster_build_mode is None else overlap_cluster_build_mode
            )
            oc_mode = "wolff" if overlap_cluster_mode is None else overlap_cluster_mode
            if oc_mode not in _CLUSTER_MODES:
                raise ValueError(f"unknown cluster_mode '{oc_mode}', expected 'wolff' or 'sw'")
            action = "update" if overlap_cluster_action is None else overlap_cluster_action
            if action not in _CLUSTER_ACTIONS:
                raise ValueError(f"unknown cluster action '{action}', expected 'update' or 'observe'")
            if overlap_cluster_update_interval < 1:
                raise ValueError("overlap_cluster interval must be >= 1")
            if action == "observe" and oc_mode == "wolff":
                raise ValueError("overlap_cluster_action='observe' requires overlap_cluster_mode='sw'")
            if snapshot_interval is not None:  # config.rs:213-219,231-235
                if snapshot_interval < 1 or snapshot_interval % overlap_cluster_update_interval != 0:
                    raise ValueError(
                        "snapshot_interval must be a positive multiple of overlap_cluster interval")
                if action == "observe":
                    raise ValueError(
                        "snapshot_interval is not supported with overlap_cluster_action='observe'")
                cfg.snapshot_interval = int(snapshot_interval)
            cfg.overlap_cluster_action = _CLUSTER_ACTIONS[action]
            if len(modes) > 4:
                raise NotImplementedError("at most 4 alternating overlap modes are supported")
            cfg.overlap_cluster_interval = int(overlap_cluster_update_interval)
            for k, m in enumerate(modes):
                cfg.overlap_cluster_modes[k] = m
            cfg.n_overlap_cluster_modes = len(modes)
            cfg.overlap_cluster_mode = _CLUSTER_MODES[oc_mode]
        if autocorrelation_max_lag is not None:
            cfg.autocorrelation_max_lag = int(autocorrelation_max_lag)
            cfg.autocorrelation_backend = _AUTOCORR_BACKENDS[backend]
        cfg.equilibration_diagnostic = 1 if equilibration_diagnostic else 0
        cfg.collect_cluster_stats = 1 if collect_cluster_stats else 0
        cfg.measure_overlap = 1 if measure_overlap else 0
        err = C.create_string_buffer(512)
        code = self._lib.pa_validate_sample_config(C.byref(cfg), self.n_replicas, err, len(err))
        if code != nat.PA_OK:
            _raise(code, err.value.decode())
        return cfg

    # ---- hot path ---------------------------------------------------------
    def run(self, cfg, wait=True):
        """pa_sim_sample.  The reference releases the GIL for the whole run and polls Python's signal state
        from its progress callback (src/lib.rs:304-333); a ctypes call cannot be interrupted, so on the main
        thread the call runs on a helper thread while this one stays interruptible: Ctrl-C asks the library to
        stop (pa_sim_request_interrupt: honoured within two batches of 256 queued sweeps), waits for it and
        re-raises.  The ensemble is then in the state after the last sweep that ran; sample() continues from it."""
        fn = self._lib.pa_sim_sample if wait else self._lib.pa_sim_sample_async
        if not wait or threading.current_thread() is not threading.main_thread():  # (timing runs enqueue directly)
            self._check(fn(self._handle, C.byref(cfg)))
            return
        outcome = {}
        done = threading.Event()

        def call():
            try:
                outcome["code"] = fn(self._handle, C.byref(cfg))
            finally:
                done.set()

        # (an Event, not Thread.join: on CPython 3.10 a KeyboardInterrupt inside join(timeout) marks the thread
        # as stopped while it is still running, so a following join() would return at once)
        threading.Thread(target=call, daemon=True).start()
        try:
            while not done.wait(0.2):
                pass
        except KeyboardInterrupt:
            # The native call must have returned before this frame is left (a later __del__ would free state it is
            # using), so further Ctrl-Cs only renew the request.  The request is repeated while waiting: the native
            # call clears the flag when it starts, and an interrupt that arrived between Thread.start() and that
            # store would otherwise be lost and the full run awaited.
            while True:
                try:
                    self._lib.pa_sim_request_interrupt(self._handle)
                    if done.wait(0.2):  # the call returns PA_ERR_INTERRUPTED within two batches of queued sweeps
                        break
                except KeyboardInterrupt:
                    continue
            self._lib.pa_sim_synchronize(self._handle)
            raise
        self._check(outcome["code"])

    def profile_step(self, cfg, n_sweeps):
        """({kernel name: (ms per step, launches per step)}, ms per step) of EVERY launch of a step of `cfg`'s protocol,
        cluster moves included (pa_sim_profile_step; the handle must have been made with stream_groups=1)."""
        cap = 64
        names = C.create_string_buffer(8192)
        ms = np.zeros(cap, dtype=np.float64)
        launches = np.zeros(cap, dtype=np.float64)
        n, step = C.c_uint64(), C.c_double()
        self._check(self._lib.pa_sim_profile_step(self._handle, C.byref(cfg), int(n_sweeps), names, len(names), nat.ptr(ms),
                                                  nat.ptr(launches), cap, C.byref(n), C.byref(step)))
        rows = names.value.decode().splitlines()
        return {rows[k]: (float(ms[k]), float(launches[k])) for k in range(int(n.value))}, float(step.value)

    def synchronize(self):
        self._check(self._lib.pa_sim_synchronize(self._handle))

    def metropolis_sweep(self, n_sweeps=1):
        self._check(self._lib.pa_sim_metropolis_sweep(self._handle, int(n_sweeps)))

    def gibbs_sweep(self, n_sweeps=1):
        self._check(self._lib.pa_sim_gibbs_sweep(self._handle, int(n_sweeps)))

    def compute_energies(self):
        self._check(self._lib.pa_sim_compute_energies(self._handle))

    def parallel_tempering(self, schedule="single_random_edge"):
        if schedule not in _PT_SCHEDULES:
            raise ValueError(
                f"unknown pt_schedule '{schedule}', expected 'single_random_edge' or 'full_ladder'"
            )
        self._check(self._lib.pa_sim_parallel_tempering(self._handle, _PT_SCHEDULES[schedule]))

    def overlap_dots(self):
        n_pairs = self.n_replicas // 2
        shape = (self.n_realizations, n_pairs, self.n_temps)
        dot_spin = np.zeros(shape, dtype=np.int64)
        dot_link = np.zeros(shape, dtype=np.int64)
        self._check(self._lib.pa_sim_overlap_dots(self._handle, nat.ptr(dot_spin), nat.ptr(dot_link)))
        return dot_spin, dot_link

    # ---- state ------------------------------------------------------------
    def get_spins(self, realization=0):
        out = np.empty(self.n_systems * self.n_spins, dtype=np.int8)
        self._check(self._lib.pa_sim_get_spins(self._handle, realization, nat.ptr(out)))
        return out

    def get_system_ids(self, realization=0):
        out = np.empty(self.n_systems, dtype=np.uint64)
        self._check(self._lib.pa_sim_get_system_ids(self._handle, realization, nat.ptr(out)))
        return out

    def get_energies(self, realization=0):
        out = np.empty(self.n_systems, dtype=np.float32)
        self._check(self._lib.pa_sim_get_energies(self._handle, realization, nat.ptr(out)))
        return out

    def get_magnetization_sums(self, realization=0):
        out = np.empty(self.n_systems, dtype=np.int64)
        self._check(self._lib.pa_sim_get_magnetization_sums(self._handle, realization, nat.ptr(out)))
        return out

    def get_rng_state(self, realization, system):
        out = np.empty(4, dtype=np.uint64)
        self._check(self._lib.pa_sim_get_rng_state(self._handle, realization, system, nat.ptr(out)))
        return out

    # setters: the reference's kernels borrow a Realization's slices (sweep.rs:220-229), so a caller may arrange them
    def set_spins(self, spins, realization=0):
        arr = np.ascontiguousarray(spins, dtype=np.int8).reshape(-1)
        if arr.size != self.n_systems * self.n_spins:
            raise ValueError(f"spins must hold {self.n_systems} x {self.n_spins} values")
        self._check(self._lib.pa_sim_set_spins(self._handle, realization, nat.ptr(arr)))

    def set_system_ids(self, system_ids, realization=0):
        arr = np.ascontiguousarray(system_ids, dtype=np.uint64).reshape(-1)
        if arr.size != self.n_systems:
            raise ValueError(f"system_ids must hold {self.n_systems} entries")
        self._check(self._lib.pa_sim_set_system_ids(self._handle, realization, nat.ptr(arr)))

    def set_rng_state(self, realization, system, state):
        arr = np.ascontiguousarray(state, dtype=np.uint64).reshape(-1)
        if arr.size != 4:
            raise ValueError("a xoshiro256** state has four 64-bit words")
        self._check(self._lib.pa_sim_set_rng_state(self._handle, realization, system, nat.ptr(arr)))

    def get_pt_counters(self, realization=0):
        n_edges = max(self.n_temps - 1, 0)
        attempts = np.zeros(max(n_edges, 1), dtype=np.uint64)
        acceptances = np.zeros(max(n_edges, 1), dtype=np.uint64)
        trips = np.zeros(self.n_systems, dtype=np.uint64)
        self._check(self._lib.pa_sim_get_pt_counters(
            self._handle, realization, nat.ptr(attempts), nat.ptr(acceptances), nat.ptr(trips)))
        return attempts[:n_edges], acceptances[:n_edges], trips.reshape(self.n_replicas, self.n_temps)

    def reset(self, seed=None):
        self._check(self._lib.pa_sim_reset(
            self._handle, self.constructor_seed if seed is None else int(seed)))

    # ---- results ----------------------------------------------------------
    def result_blocks(self):
        """[n_realizations][block] f64: the per-realisation moments (and overlap moments)."""
        length = int(self._lib.pa_sim_result_block_len(self._handle))
        out = np.empty((self.n_realizations, length), dtype=np.float64)
        self._check(self._lib.pa_sim_get_result_blocks(self._handle, nat.ptr(out)))
        return out

    def result_histograms(self):
        n_bins = self.n_spins + 1
        shape = (self.n_realizations, self.n_temps, n_bins)
        hist = np.empty(shape, dtype=np.uint64)
        qlq = np.empty(shape, dtype=np.float64)
        ql2q = np.empty(shape, dtype=np.float64)
        for r in range(self.n_realizations):
            self._check(self._lib.pa_sim_get_result_histograms(
                self._handle, r, nat.ptr(hist[r]), nat.ptr(qlq[r]), nat.ptr(ql2q[r])))
        return hist, qlq, ql2q

    def pt_diagnostics(self):
        att, acc, trips = zip(*(self.get_pt_counters(r) for r in range(self.n_realizations)))
        return {
            "edge_attempts": np.stack(att),
            "edge_acceptances": np.stack(acc),
            "round_trips": np.stack(trips),
        }

    def sample(self, n_sweeps, sweep_mode, **kwargs):
        """Runs the sweeps and returns the reference's result dictionary (src/lib.rs:337-617)."""
        cfg = self.make_config(n_sweeps, sweep_mode, **kwargs)
        self.run(cfg)
        parts, meta = self.collect_parts(cfg, kwargs.get("pt_interval") is not None)
        return assemble_parts(self._lib, parts, meta)

    # ---- per-realisation results: what a rank contributes to the disorder aggregation --------------
    def collect_parts(self, cfg, with_pt):
        """Everything the last sample() produced, BEFORE the aggregation over disorder realisations:
        `parts` maps a name to an array whose first axis is the realisation (cluster snapshots: the
        snapshot), `meta` holds what is the same for every shard of one run.  One handle assembles its
        own parts (`sample`); a disorder ensemble sharded over several handles / GPUs concatenates the
        parts in realisation order first (peapods_amd.distributed.sample) -- the aggregation code and
        its operand order are the same, so the result is bit-identical."""
        n_t, n_bins, n_real = self.n_temps, self.n_spins + 1, self.n_realizations
        parts = {"blocks": self.result_blocks()}
        meta = {"n_temps": n_t, "n_pairs": self.n_replicas // 2, "n_modes": 0, "cluster_flags": 0}
        if self.n_replicas >= 2:
            parts["hist"], parts["qlq"], parts["ql2q"] = self.result_histograms()
        if with_pt:
            for key, arr in self.pt_diagnostics().items():
                parts["pt_" + key] = arr
        if cfg.autocorrelation_max_lag:
            m2 = np.zeros((n_real, n_t), dtype=np.float64)
            q2 = np.zeros_like(m2)
            for r in range(n_real):
                self._check(self._lib.pa_sim_get_autocorrelation_tau(self._handle, r, nat.ptr(m2[r]), nat.ptr(q2[r])))
            parts["tau_mags2"] = m2
            if self.n_replicas >= 2:
                parts["tau_overlap2"] = q2
        if cfg.equilibration_diagnostic:
            n_ck = int(self._lib.pa_sim_n_equil_checkpoints(self._handle))
            sweeps = np.zeros(n_ck, dtype=np.uint64)
            energy = np.zeros((n_real, n_ck, n_t), dtype=np.float64)
            link = np.zeros_like(energy)
            for r in range(n_real):
                self._check(self._lib.pa_sim_get_equilibration(self._handle, r, nat.ptr(sweeps),
                                                               nat.ptr(energy[r]), nat.ptr(link[r])))
            meta["equil_sweeps"] = sweeps
            parts["equil_energy"], parts["equil_link"] = energy, link
        observing = (cfg.cluster_interval and cfg.cluster_action) or \
                    (cfg.overlap_cluster_interval and cfg.overlap_cluster_action)
        if cfg.collect_cluster_stats or observing:
            self._collect_cluster_parts(parts, meta, int(cfg.n_overlap_cluster_modes))
        if cfg.snapshot_interval and cfg.overlap_cluster_interval:
            self._collect_snapshot_parts(parts)
        return parts, meta

    def _collect_cluster_parts(self, parts, meta, n_modes):
        n_t, n_bins, n_real = self.n_temps, self.n_spins + 1, self.n_realizations
        modes = max(n_modes, 1)
        fk = np.zeros((n_real, n_t, n_bins), dtype=np.uint64)
        ov = np.zeros((n_real, modes, n_t, n_bins), dtype=np.uint64)
        tops = np.zeros((n_real, modes, n_t, 4), dtype=np.float64)
        got = 0
        for r in range(n_real):
            got = self._lib.pa_sim_get_cluster_stats(self._handle, r, nat.ptr(fk[r]), nat.ptr(ov[r]),
                                                     nat.ptr(tops[r]))
            if got < 0:
                self._check(got)
        meta["n_modes"], meta["cluster_flags"] = modes, int(got)
        if got & 1:
            parts["cs_fk"] = fk
        if got & 2:
            parts["cs_ov"], parts["cs_tops"] = ov, tops
        for kind, name in enumerate(_OBSERVATION_KINDS):
            rows = []
            for r in range(n_real):
                count = np.zeros(n_t, dtype=np.uint64)
                sizes = np.zeros((n_t, n_bins), dtype=np.uint64)
                top4 = np.zeros((n_t, 4), dtype=np.float64)
                density = np.zeros(n_t, dtype=np.float64)
                large = np.zeros(n_t, dtype=np.float64)
                winding = np.zeros((n_t, 4), dtype=np.float64)
                got = self._lib.pa_sim_get_observation(
                    self._handle, r, kind, nat.ptr(count), nat.ptr(sizes), nat.ptr(top4), nat.ptr(density),
                    nat.ptr(large), nat.ptr(winding))
                if got < 0:
                    self._check(got)
                if not got:
                    break
                row = {"observation_count": count, "cluster_size_counts": sizes,
                       "top_four_component_fractions": top4, "active_bond_density": density,
                       "large_component_count": large}
                if got & 2:  # src/lib.rs:84-99
                    for k, key in enumerate(("winding_x", "winding_y", "winding_either", "winding_both")):
                        row[key] = winding[:, k].copy()
                rows.append(row)
            if len(rows) == n_real and rows:  # a kind is reported when every realisation has it (src/lib.rs:447-453)
                for field in rows[0]:
                    parts[f"obs/{name}/{field}"] = np.stack([row[field] for row in rows])

    def _collect_snapshot_parts(self, parts):
        """`cluster_snapshots` of the last sample() (src/lib.rs:576-614): global realisation 0 only, so a
        handle holding a later shard contributes no rows."""
        n_t, n = self.n_temps, self.n_spins
        count = int(self._lib.pa_sim_n_snapshots(self._handle))
        head = np.zeros((count, 3), dtype=np.int64)  # sweep_id, mode_idx, has_blue
        ids = np.zeros((count, n_t, n), dtype=np.uint32)
        blue = np.zeros((count, n_t, n), dtype=np.uint32)
        spins = np.zeros((count, n_t, 2, n), dtype=np.int8)
        sids = np.zeros((count, n_t, 2), dtype=np.uint64)
        for k in range(count):
            sweep_id, mode_idx = C.c_uint64(), C.c_int32()
            has_blue = self._lib.pa_sim_get_snapshot(self._handle, k, C.byref(sweep_id), C.byref(mode_idx),
                                                     nat.ptr(ids[k]), nat.ptr(blue[k]), nat.ptr(spins[k]),
                                                     nat.ptr(sids[k]))
            if has_blue < 0:
                self._check(has_blue)
            head[k] = (int(sweep_id.value), int(mode_idx.value), int(has_blue))
        parts["snap_head"], parts["snap_ids"], parts["snap_blue"] = head, ids, blue
        parts["snap_spins"], parts["snap_sids"] = spins, sids

    def cluster_snapshots(self):
        """`cluster_snapshots` of the last sample() as the reference's list of dictionaries."""
        parts = {}
        self._collect_snapshot_parts(parts)
        return _snapshots_from_parts(parts)



_OBSERVATION_KINDS = ("fk", "houdayer", "jorg", "cmr_blue")


def _snapshots_from_parts(parts):
    snaps = []
    for k in range(parts["snap_head"].shape[0]):
        sweep_id, mode_idx, has_blue = (int(x) for x in parts["snap_head"][k])
        snap = {"sweep_id": sweep_id, "mode_idx": mode_idx, "cluster_ids": parts["snap_ids"][k],
                "spins": parts["snap_spins"][k], "system_ids": parts["snap_sids"][k]}
        if has_blue:
            snap["blue_ids"] = parts["snap_blue"][k]
        snaps.append(snap)
    return snaps


def _mean_over_realizations(lib, arr):
    """Sequential sum in realisation order, one division (results.rs:165-196,262-282)."""
    arr = np.ascontiguousarray(arr, dtype=np.float64)
    out = np.empty(arr.shape[1:], dtype=np.float64)
    lib.pa_aggregate_mean(nat.ptr(arr), arr.shape[0], int(np.prod(arr.shape[1:], dtype=np.int64)), nat.ptr(out))
    return out


def assemble_parts(lib, parts, meta):
    """The reference's result dictionary (src/lib.rs:337-617) from per-realisation parts in realisation
    order: disorder aggregation as statistics/results.rs:165-282, overlap.rs:106-152, mod.rs:905-921."""
    n_t = meta["n_temps"]
    histograms = (parts["hist"], parts["qlq"], parts["ql2q"]) if "hist" in parts else None
    pt = {k[3:]: v for k, v in parts.items() if k.startswith("pt_")} or None
    out = assemble_results(lib, parts["blocks"], histograms, pt, n_t, meta["n_pairs"])
    for key, name in (("tau_mags2", "mags2_tau"), ("tau_overlap2", "overlap2_tau")):  # results.rs:213-228,267-274
        if key in parts:
            out[name] = _mean_over_realizations(lib, parts[key])
    if "equil_sweeps" in meta:  # results.rs:229-244,275-282; src/lib.rs:559-574
        out["equil_sweeps"] = meta["equil_sweeps"]
        out["equil_energy_avg"] = _mean_over_realizations(lib, parts["equil_energy"])
        out["equil_link_overlap_avg"] = _mean_over_realizations(lib, parts["equil_link"])
    modes = meta["n_modes"]
    if "cs_fk" in parts:  # results.rs:181-214,261-267; src/lib.rs:492-543
        fk = np.ascontiguousarray(parts["cs_fk"])
        total = np.empty(fk.shape[1:], dtype=np.uint64)
        lib.pa_aggregate_sum_u64(nat.ptr(fk), fk.shape[0], fk[0].size, nat.ptr(total))
        if total.any():  # src/lib.rs:492-505
            out["fk_csd"] = [total[t].copy() for t in range(n_t)]
    if "cs_ov" in parts:
        ov = np.ascontiguousarray(parts["cs_ov"])
        total = np.empty(ov.shape[1:], dtype=np.uint64)
        lib.pa_aggregate_sum_u64(nat.ptr(ov), ov.shape[0], ov[0].size, nat.ptr(total))
        if total.any():
            out["overlap_csd"] = [[total[m, t].copy() for t in range(n_t)] for m in range(modes)]
        mean = _mean_over_realizations(lib, parts["cs_tops"])
        out["top_cluster_sizes"] = [mean[m].copy() for m in range(modes)]
    observations = {}
    for key, arr in parts.items():
        if key.startswith("obs/"):
            _, kind, field = key.split("/")
            observations.setdefault(kind, {})[field] = arr
    if observations:  # src/lib.rs:414-456
        ordered = {kind: observations[kind] for kind in _OBSERVATION_KINDS if kind in observations}
        out.setdefault("per_disorder", {})["cluster_observations"] = ordered
    if "snap_head" in parts:
        snaps = _snapshots_from_parts(parts)
        if snaps:
            out["cluster_snapshots"] = snaps
    return out


def assemble_results(lib, blocks, histograms, pt, n_temps, n_pairs):
    """Disorder aggregation (statistics/results.rs:165-282, overlap.rs:106-152, mod.rs:905-921)
    of per-realisation blocks, in realisation order, into the reference's dictionary."""
    n_real = blocks.shape[0]
    blocks = np.ascontiguousarray(blocks)
    mean = np.empty(blocks.shape[1], dtype=np.float64)
    lib.pa_aggregate_mean(nat.ptr(blocks), n_real, blocks.shape[1], nat.ptr(mean))
    out = {}
    for k, key in enumerate(["mags", "mags2", "mags4", "energies", "energies2"]):
        out[key] = mean[k * n_temps:(k + 1) * n_temps].copy()
    if n_pairs > 0:
        keys = ["overlap", "overlap2", "overlap4", "link_overlap", "link_overlap2", "link_overlap4"]
        for k, key in enumerate(keys):
            out[key] = mean[(5 + k) * n_temps:(6 + k) * n_temps].copy()
        hist, qlq, ql2q = (np.ascontiguousarray(a) for a in histograms)
        per = hist[0].size
        hist_sum = np.empty(hist.shape[1:], dtype=np.uint64)
        qlq_sum = np.empty(qlq.shape[1:], dtype=np.float64)
        ql2q_sum = np.empty(ql2q.shape[1:], dtype=np.float64)
        lib.pa_aggregate_sum_u64(nat.ptr(hist), n_real, per, nat.ptr(hist_sum))
        lib.pa_aggregate_sum_f64(nat.ptr(qlq), n_real, per, nat.ptr(qlq_sum))
        lib.pa_aggregate_sum_f64(nat.ptr(ql2q), n_real, per, nat.ptr(ql2q_sum))
        out["overlap_histogram"] = [hist_sum[t].copy() for t in range(n_temps)]
        out["ql_at_q_sum"] = qlq_sum
        out["ql2_at_q_sum"] = ql2q_sum
        if n_real > 1:
            out["per_sample_overlap_histogram"] = hist
            out["per_sample_ql_at_q_sum"] = qlq
            out["per_sample_ql2_at_q_sum"] = ql2q
    if pt is not None:
        out["per_disorder"] = {"parallel_tempering": pt}
    return out
