"""`Ising`: user-facing model class, a drop-in for `peapods.Ising` on the hot path
(reference python/peapods/spin_models.py).  Seeding, coupling generation and the
derived observables follow the reference so that a script written against
`peapods.Ising` runs unchanged; the Monte Carlo work happens on the MI355X.
"""

import numpy as np

from ._core import IsingSimulation

# reference spin_models.py:5-10
GEOMETRIES = {
    "triangular": [[1, 0], [0, 1], [1, -1]],
    "tri": [[1, 0], [0, 1], [1, -1]],
    "fcc": [[1, 1, 0], [1, 0, 1], [0, 1, 1], [1, -1, 0], [1, 0, -1], [0, 1, -1]],
    "bcc": [[1, 1, 1], [1, 1, -1], [1, -1, 1], [1, -1, -1]],
}

_DEFAULT_TEMPERATURES = np.geomspace(0.1, 10, 32)


def seed_material(seed):
    """(coupling SeedSequence, dynamics root u64) from one user seed (spin_models.py:13-19)."""
    if seed is not None and (not isinstance(seed, (int, np.integer)) or seed < 0):
        raise ValueError("seed must be a non-negative integer or None")
    coupling_seq, dynamics_seq = np.random.SeedSequence(seed).spawn(2)
    return coupling_seq, int(dynamics_seq.generate_state(1, dtype=np.uint64)[0])


def generate_couplings(kind, lattice_shape, n_neighbors, n_disorder, coupling_seq,
                       realizations=None):
    """Built-in coupling ensembles (spin_models.py:104-126).  Each disorder sample draws from
    its own child of the coupling SeedSequence, so sample k does not depend on n_disorder.
    `realizations` selects a subset (used to shard disorder over GPUs)."""
    single = tuple(lattice_shape) + (n_neighbors,)
    children = coupling_seq.spawn(n_disorder)
    wanted = range(n_disorder) if realizations is None else realizations
    out = []
    for index in wanted:
        rng = np.random.default_rng(children[index])
        if kind == "ferro":
            sample = np.ones(single, dtype=np.float32)
        elif kind == "bimodal":
            sample = (2 * rng.integers(0, 2, size=single) - 1).astype(np.float32)
        elif kind == "gaussian":
            sample = rng.standard_normal(single).astype(np.float32)
        else:
            raise ValueError("Invalid mode for couplings.")
        out.append(sample)
    return out


class Ising:
    """Ising / spin-glass model on a periodic lattice, sampled on the GPU.

    Same constructor and `sample()` surface as `peapods.Ising`.
    """

    def __init__(self, lattice_shape, couplings="ferro", temperatures=_DEFAULT_TEMPERATURES,
                 n_replicas=1, n_disorder=1, neighbor_offsets=None, geometry=None, seed=None,
                 device=0, comm=None):
        """`comm` (not in the reference, which is one process): an exchange of peapods_amd.distributed (`Comm` =
        RCCL, one rank per GPU).  The n_disorder realisations are then sharded over its ranks -- this rank
        builds and runs only its own -- and `sample()` returns, on every rank, the result over all of them,
        bit-identical to a single-GPU run of the same arguments."""
        if geometry is not None:
            if neighbor_offsets is not None:
                raise ValueError("Cannot specify both geometry and neighbor_offsets")
            if geometry not in GEOMETRIES:
                raise ValueError(
                    f"Unknown geometry '{geometry}', choose from: {list(GEOMETRIES.keys())}"
                )
            neighbor_offsets = GEOMETRIES[geometry]

        self.lattice_shape = tuple(lattice_shape)
        self.n_spins = int(np.prod(lattice_shape))
        self.n_dims = len(lattice_shape)
        self.n_neighbors = len(neighbor_offsets) if neighbor_offsets else self.n_dims
        self.temperatures = np.asarray(temperatures).copy().astype(np.float32)
        self.n_temps = len(temperatures)
        self.n_replicas = n_replicas
        self.n_disorder = n_disorder
        self.seed = seed
        coupling_seq, self._constructor_dynamics_seed = seed_material(seed)

        self._comm = comm
        first, stop = 0, n_disorder
        if comm is not None:
            from .distributed import shard_range

            if n_disorder < comm.world:
                raise ValueError(f"n_disorder = {n_disorder} realisations cannot be sharded over {comm.world} ranks")
            first, stop = shard_range(n_disorder, comm.rank, comm.world)
        if isinstance(couplings, np.ndarray):
            coup = couplings.astype(np.float32)
            if comm is not None:
                if n_disorder == 1 and coup.ndim == len(self.lattice_shape) + 1:
                    coup = coup[None]  # (the un-batched shape, accepted without comm too)
                if coup.ndim != len(self.lattice_shape) + 2 or coup.shape[0] != n_disorder:
                    raise ValueError("with comm, explicit couplings must have shape (n_disorder, *lattice, z')")
                coup = coup[first:stop]
        else:
            samples = generate_couplings(couplings, self.lattice_shape, self.n_neighbors,
                                         n_disorder, coupling_seq, realizations=range(first, stop))
            coup = samples[0] if n_disorder == 1 else np.stack(samples)
        self.couplings = coup  # (with comm: this rank's shard)
        self._sim = IsingSimulation(list(lattice_shape), coup, self.temperatures, n_replicas,
                                    neighbor_offsets, self._constructor_dynamics_seed,
                                    device=device, first_realization=first)

    def reset(self, seed=None):
        """Replay the constructor's dynamics, or a one-off seeded reset; couplings stay."""
        self._sim.reset(None if seed is None else seed_material(seed)[1])

    def sample(self, n_sweeps, sweep_mode="metropolis", cluster_update_interval=None,
               cluster_mode="sw", cluster_action="update", pt_interval=None,
               pt_schedule="single_random_edge", overlap_cluster_update_interval=None,
               overlap_cluster_build_mode="houdayer", overlap_cluster_mode="wolff",
               overlap_cluster_action="update", warmup_ratio=0.25, collect_cluster_stats=False,
               autocorrelation_max_lag=None, autocorrelation_backend="ring", sequential=False,
               equilibration_diagnostic=False, snapshot_interval=None):
        """Run the sweeps and set the derived observables (reference spin_models.py:146-337)."""
        # the reference's Python-side checks, same messages (spin_models.py:222-247)
        if cluster_action not in {"update", "observe"}:
            raise ValueError("cluster_action must be 'update' or 'observe'")
        if overlap_cluster_action not in {"update", "observe"}:
            raise ValueError("overlap_cluster_action must be 'update' or 'observe'")
        if pt_schedule not in {"single_random_edge", "full_ladder"}:
            raise ValueError("pt_schedule must be 'single_random_edge' or 'full_ladder'")
        if autocorrelation_backend not in {"ring", "fft"}:
            raise ValueError("autocorrelation_backend must be 'ring' or 'fft'")
        if autocorrelation_backend == "fft" and autocorrelation_max_lag is None:
            raise ValueError("autocorrelation_backend='fft' requires autocorrelation_max_lag")
        if cluster_action == "observe" and cluster_update_interval is None:
            raise ValueError("cluster_action='observe' requires cluster_update_interval")
        if overlap_cluster_action == "observe" and overlap_cluster_update_interval is None:
            raise ValueError(
                "overlap_cluster_action='observe' requires overlap_cluster_update_interval"
            )

        oci = overlap_cluster_update_interval
        if self._comm is not None:
            from functools import partial

            from . import distributed

            run = partial(distributed.sample, self._sim, self._comm)
        else:
            run = self._sim.sample
        result = run(
            n_sweeps,
            sweep_mode,
            cluster_update_interval=cluster_update_interval,
            cluster_mode=cluster_mode if cluster_update_interval else None,
            cluster_action=cluster_action if cluster_update_interval else None,
            pt_interval=pt_interval,
            pt_schedule=pt_schedule,
            overlap_cluster_update_interval=oci,
            overlap_cluster_build_mode=overlap_cluster_build_mode if oci else None,
            overlap_cluster_mode=overlap_cluster_mode if oci else None,
            overlap_cluster_action=overlap_cluster_action if oci else None,
            warmup_ratio=warmup_ratio,
            collect_cluster_stats=collect_cluster_stats,
            autocorrelation_max_lag=autocorrelation_max_lag,
            autocorrelation_backend=autocorrelation_backend,
            sequential=sequential,
            equilibration_diagnostic=equilibration_diagnostic,
            snapshot_interval=snapshot_interval if oci else None,
        )
        self._set_observables(result)
        return result

    def _set_observables(self, result):
        self.mags = result["mags"]
        self.mags2 = result["mags2"]
        self.mags4 = result["mags4"]
        self.energies_avg = result["energies"]
        self.energies2_avg = result["energies2"]
        # spin_models.py:276-281
        self.binder_cumulant = 1 - self.mags4 / (3 * self.mags2**2)
        self.heat_capacity = (
            self.n_spins * (self.energies2_avg - self.energies_avg**2) / self.temperatures**2
        )
        if "overlap2" in result:  # spin_models.py:283-293
            self.overlap = result["overlap"]
            self.overlap2 = result["overlap2"]
            self.overlap4 = result["overlap4"]
            self.sg_binder = 1 - self.overlap4 / (3 * self.overlap2**2)
            self.link_overlap = result["link_overlap"]
            self.link_overlap2 = result["link_overlap2"]
            self.link_overlap4 = result["link_overlap4"]
            self.link_overlap_binder = 1 - self.link_overlap4 / (3 * self.link_overlap2**2)
        for key in ("overlap_histogram", "ql_at_q_sum", "ql2_at_q_sum",
                    "per_sample_overlap_histogram", "per_sample_ql_at_q_sum",
                    "per_sample_ql2_at_q_sum"):
            if key in result:
                setattr(self, key, result[key])
        if "fk_csd" in result:  # spin_models.py:309-317: mean cluster size <s^2>/<s> per temperature
            self.fk_csd = result["fk_csd"]
            mean_size = np.empty(self.n_temps)
            for t, hist in enumerate(self.fk_csd):
                sizes = np.arange(len(hist))
                weighted = sizes * hist
                n_sites = weighted.sum()
                mean_size[t] = (sizes * weighted).sum() / n_sites if n_sites > 0 else 0.0
            self.mean_cluster_size = mean_size
        for key in ("top_cluster_sizes", "cluster_snapshots"):  # spin_models.py:319-333
            if key in result:
                setattr(self, key, result[key])
        for key in ("mags2_tau", "overlap2_tau"):  # spin_models.py:325-328
            if key in result:
                setattr(self, key, result[key])
        if "equil_sweeps" in result:  # spin_models.py:330-333
            self._equil_sweeps = result["equil_sweeps"]
            self._equil_energy_avg = result["equil_energy_avg"]
            self._equil_link_overlap_avg = result["equil_link_overlap_avg"]
        self.per_disorder = result.get("per_disorder", {})

    def equilibration_delta(self, j_squared=1.0):
        """Delta(t) = e(t) - J^2 beta z (1 - q_l(t)) at the checkpoints of the last run with
        `equilibration_diagnostic=True` (spin_models.py:339-358; the engine's energy is
        +sum J s s / N, so the sign is that of the reference).  Returns (sweeps, delta[n_ckpt, n_temps])."""
        beta = 1.0 / self.temperatures
        delta = self._equil_energy_avg - j_squared * beta * self.n_neighbors * (
            1 - self._equil_link_overlap_avg
        )
        return self._equil_sweeps, delta

    def get_energies(self):
        """Mean energies per temperature from the last `sample` run."""
        return self.energies_avg
