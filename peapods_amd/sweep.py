"""Multi-size / multi-configuration runs and the `.npz` result files.

Host-side mirror of the reference's `peapods.sweep` (python/peapods/sweep.py): `run_sweep` builds one
`Ising` per (coupling, overlap build mode, overlap cluster mode, lattice size), runs `sample()` on
the GPU engine and optionally writes `sweep_<config>.npz` with the reference's key names
(sweep.py:100-163), so scripts that read those files work unchanged.  Plots are drawn only when
matplotlib is importable (it is not part of this image).
"""

import itertools
import sys
import time
from pathlib import Path

import numpy as np

from .spin_models import Ising

# spawn-key tags of the per-run seed tree (sweep.py:10)
COUPLING_TAG = {"ferro": 0, "bimodal": 1, "gaussian": 2}

# model attribute -> suffix of the npz key, in the order the reference writes them (sweep.py:105-135)
_OPTIONAL_ARRAYS = (
    ("sg_binder", "sg_binder"),
    ("mean_cluster_size", "mean_cluster_size"),
    ("top_cluster_sizes", "top_cluster_sizes"),
)
_SNAPSHOT_FIELDS = (("cluster_ids", "snapshot_cluster_ids"), ("spins", "snapshot_spins"),
                    ("system_ids", "snapshot_system_ids"))


def run_seed_words(seed):
    """Four u32 words of `SeedSequence(seed)`: the root of a sweep's seed tree (sweep.py:13-19)."""
    if seed is not None:
        if not isinstance(seed, (int, np.integer)) or seed < 0:
            raise ValueError("seed must be a non-negative integer or None")
    words = np.random.SeedSequence(seed).generate_state(4, dtype=np.uint32)
    return [int(w) for w in words]


def run_child_seed(root_words, coupling, shape):
    """Seed of one (coupling, shape) run: child (tag, ndim, *shape) of the root (sweep.py:22-27)."""
    key = (COUPLING_TAG[coupling], len(shape)) + tuple(int(s) for s in shape)
    child = np.random.SeedSequence(root_words, spawn_key=key)
    return int(child.generate_state(1, dtype=np.uint64)[0])


def flatten_per_disorder(per_disorder, prefix=""):
    """`per_disorder` nested dict -> flat npz keys (sweep.py:30-43)."""
    head = f"{prefix}_" if prefix else ""
    out = {}
    for kind, fields in per_disorder.get("cluster_observations", {}).items():
        for name, values in fields.items():
            out[f"{head}per_disorder_cluster_observations_{kind}_{name}"] = values
    for name, values in (per_disorder.get("parallel_tempering") or {}).items():
        out[f"{head}per_disorder_pt_{name}"] = values
    return out


def cumulative_overlap_ratio(per_sample_hist):
    """I(q)/X(q) of Billoire et al. from per-sample overlap histograms (sweep.py:46-75).

    per_sample_hist [n_disorder, n_temps, n_bins] -> (q >= 0 grid, median/mean ratio [n_temps, n_q],
    mean X, median X), X(q) = weight of |q'| <= q in one sample.
    """
    hist = np.asarray(per_sample_hist)
    n_disorder, n_temps, n_bins = hist.shape
    centre = n_bins // 2
    q_grid = np.linspace(-1, 1, n_bins)[centre:]
    x = np.zeros((n_disorder, n_temps, len(q_grid)))
    for k in range(len(q_grid)):
        x[:, :, k] = hist[:, :, centre - k:centre + k + 1].sum(axis=2)
    norm = hist.sum(axis=2, keepdims=True)
    x /= np.where(norm == 0, 1, norm)
    mean = x.mean(axis=0)
    median = np.median(x, axis=0)
    ratio = np.where(mean > 0, median / np.where(mean > 0, mean, 1), 0.0)
    return q_grid, ratio, mean, median


def config_label(coupling, build_mode, oc_mode):
    """File / legend label of one configuration (sweep.py:78-84)."""
    label = coupling
    if build_mode != "houdayer":
        label += f"_{build_mode}"
    if oc_mode != "wolff":
        label += f"_{oc_mode}"
    return label


def size_label(shape):
    return "x".join(str(s) for s in shape)


def model_arrays(model, prefix):
    """The arrays one model contributes to a sweep file, keyed as the reference does."""
    d = {
        f"{prefix}_lattice_shape": np.array(model.lattice_shape),
        f"{prefix}_binder_cumulant": model.binder_cumulant,
        f"{prefix}_heat_capacity": model.heat_capacity,
        f"{prefix}_energies": model.energies_avg,
    }
    for attr, suffix in _OPTIONAL_ARRAYS:
        if hasattr(model, attr):
            d[f"{prefix}_{suffix}"] = getattr(model, attr)
    if hasattr(model, "overlap_histogram"):
        d[f"{prefix}_overlap_histogram"] = np.array(list(model.overlap_histogram))
    if hasattr(model, "per_sample_overlap_histogram"):
        psh = model.per_sample_overlap_histogram
        d[f"{prefix}_per_sample_overlap_histogram"] = psh
        q_grid, ratio, _, _ = cumulative_overlap_ratio(psh)
        d[f"{prefix}_cumulative_overlap_q"] = q_grid
        d[f"{prefix}_cumulative_overlap_ratio"] = ratio
    for attr in ("mags2_tau", "overlap2_tau"):
        if hasattr(model, attr):
            d[f"{prefix}_{attr}"] = getattr(model, attr)
    if hasattr(model, "_equil_sweeps"):
        d[f"{prefix}_equil_sweeps"] = model._equil_sweeps
        d[f"{prefix}_equil_energy_avg"] = model._equil_energy_avg
        d[f"{prefix}_equil_link_overlap_avg"] = model._equil_link_overlap_avg
    snaps = getattr(model, "cluster_snapshots", None)
    if snaps:
        d[f"{prefix}_snapshot_sweep_ids"] = np.array([s["sweep_id"] for s in snaps], dtype=np.int64)
        d[f"{prefix}_snapshot_mode_idxs"] = np.array([s["mode_idx"] for s in snaps], dtype=np.int64)
        for field, suffix in _SNAPSHOT_FIELDS:
            d[f"{prefix}_{suffix}"] = np.stack([s[field] for s in snaps])
        if "blue_ids" in snaps[0]:
            d[f"{prefix}_snapshot_blue_ids"] = np.stack([s["blue_ids"] for s in snaps])
    d.update(flatten_per_disorder(model.per_disorder, prefix=prefix))
    return d


def write_sweep_file(models, label, temperatures, output_dir):
    """Write `sweep_<label>.npz` for the sizes of one configuration (sweep.py:100-163)."""
    arrays = {"temperatures": temperatures}
    for slabel, model in models.items():
        arrays.update(model_arrays(model, slabel))
    path = Path(output_dir) / f"sweep_{label}.npz"
    np.savez(path, **arrays)
    print(f"  Data saved to {path}")
    return path


def simulate_arrays(model, result):
    """The arrays `peapods simulate -o` stores (cli.py:608-639)."""
    d = {
        "temperatures": model.temperatures,
        "binder_cumulant": model.binder_cumulant,
        "heat_capacity": model.heat_capacity,
    }
    for key in ("mags", "mags2", "mags4", "energies", "energies2",
                "overlap", "overlap2", "overlap4"):
        if key in result:
            d[key] = result[key]
    for attr in ("sg_binder", "mean_cluster_size", "fk_csd", "top_cluster_sizes",
                 "per_sample_overlap_histogram"):
        if hasattr(model, attr):
            d[attr] = getattr(model, attr)
    d.update(flatten_per_disorder(model.per_disorder))
    return d


def _plots(models, label, temperatures, output_dir, collect_cluster_stats):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    def curve(fname, ylabel, pick, logx=True):
        fig, ax = plt.subplots(figsize=(6, 4))
        for slabel, model in models.items():
            ax.plot(temperatures, pick(model), label=slabel)
        ax.set_xlabel("Temperature")
        ax.set_ylabel(ylabel)
        if logx:
            ax.set_xscale("log")
        ax.legend()
        ax.set_title(label)
        path = Path(output_dir) / fname
        fig.savefig(path, dpi=150, bbox_inches="tight")
        plt.close(fig)
        print(f"  Plot saved to {path}")

    glass = any(hasattr(m, "sg_binder") for m in models.values())
    curve(f"binder_{label}.png", "SG Binder" if glass else "Binder cumulant",
          lambda m: m.sg_binder if glass else m.binder_cumulant)
    curve(f"heat_capacity_{label}.png", "$C_v$", lambda m: m.heat_capacity, logx=False)
    for slabel, model in models.items():
        if hasattr(model, "overlap_histogram"):
            hist = np.array(list(model.overlap_histogram), dtype=np.float64)
            q = np.linspace(-1, 1, hist.shape[1])
            fig, ax = plt.subplots(figsize=(6, 4))
            for t, h in enumerate(hist):
                if h.sum() > 0:
                    ax.plot(q, h / h.sum() / (2.0 / (hist.shape[1] - 1)), alpha=0.7)
            ax.set_xlabel("$q$")
            ax.set_ylabel("$P(q)$")
            path = Path(output_dir) / f"pq_{slabel}_{label}.png"
            fig.savefig(path, dpi=150, bbox_inches="tight")
            plt.close(fig)
            print(f"  Plot saved to {path}")


def run_sweep(sizes, *, couplings=("ferro",), temperatures, n_replicas=1, n_disorder=1,
              neighbor_offsets=None, geometry=None, n_sweeps, sweep_mode="metropolis",
              cluster_update_interval=None, cluster_mode="sw", cluster_action="update",
              pt_interval=None, pt_schedule="single_random_edge",
              overlap_cluster_update_interval=None, overlap_cluster_build_modes=("houdayer",),
              overlap_cluster_modes=("wolff",), overlap_cluster_action="update",
              warmup_ratio=0.25, collect_cluster_stats=False, autocorrelation_max_lag=None,
              autocorrelation_backend="ring", autocorrelation_plot_temp=None,
              equilibration_diagnostic=False, save_plots=False, save_data=False,
              output_dir=".", sequential=False, snapshot_interval=None, seed=None, device=0):
    """Run every (configuration, size) and return `{config_label: {size_label: Ising}}`
    (sweep.py:351-512; same keywords)."""
    write_data = save_data
    if save_plots:
        try:
            import matplotlib  # noqa: F401
        except ImportError:
            print("error: matplotlib is required for --save-plots.", file=sys.stderr)
            sys.exit(1)
    if save_plots or write_data:
        Path(output_dir).mkdir(parents=True, exist_ok=True)

    configs = []
    for coupling, build_mode, oc_mode in itertools.product(
            couplings, overlap_cluster_build_modes, overlap_cluster_modes):
        # a non-default build mode without an overlap interval would silently do nothing
        if build_mode != "houdayer" and overlap_cluster_update_interval is None:
            print(f"  skip: {config_label(coupling, build_mode, oc_mode)} — "
                  f"overlap_cluster_build_mode={build_mode} set but no "
                  f"--overlap-cluster-update-interval", file=sys.stderr)
            continue
        configs.append((coupling, build_mode, oc_mode))
    n_runs = len(configs) * len(sizes)

    root = run_seed_words(seed)
    results = {}
    started = time.perf_counter()
    run = 0
    for coupling, build_mode, oc_mode in configs:
        label = config_label(coupling, build_mode, oc_mode)
        models = {}
        for shape in sizes:
            run += 1
            slabel = size_label(shape)
            print(f"[{run}/{n_runs}] {slabel}, {label}")
            model = Ising(shape, couplings=coupling, temperatures=temperatures,
                          n_replicas=n_replicas, n_disorder=n_disorder,
                          neighbor_offsets=neighbor_offsets, geometry=geometry,
                          seed=run_child_seed(root, coupling, shape), device=device)
            t0 = time.perf_counter()
            model.sample(
                n_sweeps, sweep_mode=sweep_mode,
                cluster_update_interval=cluster_update_interval, cluster_mode=cluster_mode,
                cluster_action=cluster_action, pt_interval=pt_interval, pt_schedule=pt_schedule,
                overlap_cluster_update_interval=overlap_cluster_update_interval,
                overlap_cluster_build_mode=build_mode, overlap_cluster_mode=oc_mode,
                overlap_cluster_action=overlap_cluster_action, warmup_ratio=warmup_ratio,
                collect_cluster_stats=collect_cluster_stats,
                autocorrelation_max_lag=autocorrelation_max_lag,
                autocorrelation_backend=autocorrelation_backend, sequential=sequential,
                equilibration_diagnostic=equilibration_diagnostic,
                snapshot_interval=snapshot_interval)
            print(f"  {time.perf_counter() - t0:.2f}s")
            models[slabel] = model
        results[label] = models
        if write_data:
            write_sweep_file(models, label, temperatures, output_dir)
        if save_plots:
            _plots(models, label, temperatures, output_dir, collect_cluster_stats)
    print(f"\nSweep complete: {n_runs} runs in {time.perf_counter() - started:.1f}s")
    return results
