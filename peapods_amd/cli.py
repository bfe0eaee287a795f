"""`python -m peapods_amd.cli simulate|bench|sweep ...` — the reference's command line on the GPU engine.

Same sub-commands, flags and TOML sections as `peapods` (python/peapods/cli.py): `simulate` prints the
per-temperature table and optionally stores the `.npz` (cli.py:595-640), `bench` prints wall time
per sweep (cli.py:643-653), `sweep` runs `peapods_amd.sweep.run_sweep` from flags and/or a TOML file
(cli.py:355-566).  Only the argument surface is mirrored; everything runs through `Ising.sample`.
"""

import argparse
import json
import sys
import time

import numpy as np

try:  # Python >= 3.11
    import tomllib
except ModuleNotFoundError:  # this image: Python 3.10 + tomli
    import tomli as tomllib

from . import sweep as sweep_mod
from .spin_models import Ising

COUPLINGS = ("ferro", "bimodal", "gaussian")
GEOMETRY_NAMES = ("triangular", "tri", "fcc", "bcc")

# (flag, kwargs for add_argument, default used by simulate/bench).  The sweep sub-command takes the
# same flags with default None ("not given": the TOML file or SWEEP_DEFAULTS decides).
SAMPLING_FLAGS = (
    ("--geometry", dict(choices=GEOMETRY_NAMES, help="Named lattice geometry"), None),
    ("--neighbor-offsets", dict(type=str, help="JSON list of offset vectors, e.g. '[[1,0],[0,1]]'"),
     None),
    ("--n-replicas", dict(type=int), 1),
    ("--n-disorder", dict(type=int), 1),
    ("--seed", dict(type=int), None),
    ("--temp-min", dict(type=float), "required"),
    ("--temp-max", dict(type=float), "required"),
    ("--n-temps", dict(type=int), 32),
    ("--temp-scale", dict(choices=("linear", "log"), help="Temperature spacing (default: log)"),
     "log"),
    ("--n-sweeps", dict(type=int), "required"),
    ("--sweep-mode", dict(choices=("metropolis", "gibbs")), "metropolis"),
    ("--cluster-interval", dict(type=int, help="Cluster update every N sweeps"), None),
    ("--cluster-mode", dict(choices=("sw", "wolff")), "sw"),
    ("--cluster-action", dict(choices=("update", "observe")), "update"),
    ("--pt-interval", dict(type=int, help="Parallel tempering every N sweeps"), None),
    ("--pt-schedule", dict(choices=("single_random_edge", "full_ladder")), "single_random_edge"),
    ("--overlap-cluster-update-interval",
     dict(type=int, help="Overlap cluster move every N sweeps (requires n_replicas >= 2)"), None),
    ("--autocorrelation-max-lag", dict(type=int, help="Max lag for autocorrelation of m^2 and q^2"),
     None),
    ("--autocorrelation-backend", dict(choices=("ring", "fft")), "ring"),
)
BOOLEAN_FLAGS = (
    ("--collect-cluster-stats",
     "Collect FK cluster size distribution and top-4 overlap cluster sizes"),
    ("--equilibration-diagnostic",
     "Track energy + link-overlap running averages for equilibration check"),
)

# defaults of the sweep sub-command after flags and TOML are merged (cli.py:318-352)
SWEEP_DEFAULTS = dict(
    sizes=None, couplings=("ferro",), temp_min=None, temp_max=None, n_temps=32, temp_scale="log",
    n_replicas=1, n_disorder=1, seed=None, neighbor_offsets=None, geometry=None, n_sweeps=None,
    sweep_mode="metropolis", cluster_interval=None, cluster_mode="sw", cluster_action="update",
    pt_interval=None, pt_schedule="single_random_edge", overlap_cluster_update_interval=None,
    overlap_cluster_build_mode=("houdayer",), overlap_cluster_mode=("wolff",),
    overlap_cluster_action="update", warmup_ratio=0.25, collect_cluster_stats=False,
    autocorrelation_max_lag=None, autocorrelation_backend="ring", autocorrelation_plot_temp=None,
    equilibration_diagnostic=False, save_plots=False, save_data=False, output_dir=".",
    sequential=False, snapshot_interval=None,
)

# TOML (section, key) -> keyword, with an optional conversion (cli.py:355-460)
_as_tuple = tuple
_as_list_tuple = lambda v: tuple(v if isinstance(v, list) else [v])  # noqa: E731
TOML_MAP = (
    ("lattice", "sizes", "sizes", lambda v: [tuple(s) for s in v]),
    ("lattice", "geometry", "geometry", None),
    ("lattice", "neighbor_offsets", "neighbor_offsets", lambda v: [list(o) for o in v]),
    ("lattice", "couplings", "couplings", _as_tuple),
    ("temperatures", "min", "temp_min", None),
    ("temperatures", "max", "temp_max", None),
    ("temperatures", "count", "n_temps", None),
    ("temperatures", "scale", "temp_scale", None),
    ("replicas", "n_replicas", "n_replicas", None),
    ("replicas", "n_disorder", "n_disorder", None),
    ("sampling", "n_sweeps", "n_sweeps", None),
    ("sampling", "sweep_mode", "sweep_mode", None),
    ("sampling", "warmup_ratio", "warmup_ratio", None),
    ("sampling", "sequential", "sequential", None),
    ("sampling", "seed", "seed", None),
    ("cluster", "interval", "cluster_interval", None),
    ("cluster", "mode", "cluster_mode", None),
    ("cluster", "action", "cluster_action", None),
    ("parallel_tempering", "interval", "pt_interval", None),
    ("parallel_tempering", "schedule", "pt_schedule", None),
    ("overlap_cluster", "interval", "overlap_cluster_update_interval", None),
    ("overlap_cluster", "build_modes", "overlap_cluster_build_mode", _as_tuple),
    ("overlap_cluster", "cluster_mode", "overlap_cluster_mode", _as_list_tuple),
    ("overlap_cluster", "snapshot_interval", "snapshot_interval", None),
    ("overlap_cluster", "action", "overlap_cluster_action", None),
    ("diagnostics", "collect_cluster_stats", "collect_cluster_stats", None),
    ("diagnostics", "equilibration_diagnostic", "equilibration_diagnostic", None),
    ("output", "save_plots", "save_plots", None),
    ("output", "save_data", "save_data", None),
    ("output", "dir", "output_dir", None),
)
TOML_AUTOCORR = (("max_lag", "autocorrelation_max_lag"), ("backend", "autocorrelation_backend"),
                 ("plot_temp", "autocorrelation_plot_temp"))


def _add_sampling_flags(parser, for_sweep):
    for flag, kw, default in SAMPLING_FLAGS:
        kw = dict(kw)
        if for_sweep:
            kw["default"] = None
        elif default == "required":
            kw["required"] = True
        else:
            kw["default"] = default
        parser.add_argument(flag, **kw)
    for flag, text in BOOLEAN_FLAGS:
        parser.add_argument(flag, action="store_true", default=None if for_sweep else False,
                            help=text)


def _add_model_flags(parser):
    """Flags of `simulate` and `bench` (cli.py:99-125)."""
    parser.add_argument("--shape", type=int, nargs="+", required=True,
                        help="Lattice dimensions, e.g. --shape 32 32")
    parser.add_argument("--couplings", default="ferro", choices=COUPLINGS,
                        help="Coupling distribution (default: ferro)")
    parser.add_argument("--overlap-cluster-build-mode", default="houdayer")
    parser.add_argument("--overlap-cluster-mode", default="wolff", choices=("wolff", "sw"))
    parser.add_argument("--overlap-cluster-action", default="update",
                        choices=("update", "observe"))
    _add_sampling_flags(parser, for_sweep=False)
    parser.add_argument("--device", type=int, default=0, help="GPU ordinal (this engine only)")


def _add_sweep_flags(parser):
    """Flags of `sweep` (cli.py:222-271); every default is None = "take it from the TOML file"."""
    parser.add_argument("--config", type=str, default=None, help="Path to TOML config file")
    parser.add_argument("--sizes", nargs="+", default=None,
                        help="Lattice sizes as comma-separated dims, e.g. --sizes 8,8 16,16 8,8,8")
    parser.add_argument("--couplings", nargs="+", default=None, choices=COUPLINGS,
                        help="Coupling distributions to sweep (default: ferro)")
    parser.add_argument("--overlap-cluster-build-mode", nargs="+", default=None)
    parser.add_argument("--overlap-cluster-mode", nargs="+", default=None, choices=("wolff", "sw"))
    parser.add_argument("--overlap-cluster-action", default=None, choices=("update", "observe"))
    _add_sampling_flags(parser, for_sweep=True)
    parser.add_argument("--sequential", action="store_true", default=None,
                        help="Accepted for compatibility; the GPU engine has no such mode")
    parser.add_argument("--snapshot-interval", type=int, default=None,
                        help="Save cluster snapshots every N sweeps")
    parser.add_argument("--warmup-ratio", type=float, default=None)
    parser.add_argument("--autocorrelation-plot-temp", type=float, default=None)
    parser.add_argument("--save-plots", action="store_true", default=None)
    parser.add_argument("--save-data", action="store_true", default=None, help="Save data as .npz")
    parser.add_argument("--output-dir", default=None, help="Output directory (default: .)")
    parser.add_argument("--device", type=int, default=None, help="GPU ordinal (this engine only)")


def temperature_grid(t_min, t_max, count, scale):
    """`linear` -> linspace, otherwise geomspace (cli.py:312-315)."""
    return (np.linspace if scale == "linear" else np.geomspace)(t_min, t_max, count)


def build_model(args):
    offsets = json.loads(args.neighbor_offsets) if args.neighbor_offsets is not None else None
    return Ising(tuple(args.shape), couplings=args.couplings,
                 temperatures=temperature_grid(args.temp_min, args.temp_max, args.n_temps,
                                               args.temp_scale),
                 n_replicas=args.n_replicas, n_disorder=args.n_disorder, neighbor_offsets=offsets,
                 geometry=args.geometry, seed=args.seed, device=args.device)


def sample_kwargs(args):
    """argparse namespace -> `Ising.sample` keywords (cli.py:293-309)."""
    return dict(
        sweep_mode=args.sweep_mode, cluster_update_interval=args.cluster_interval,
        cluster_mode=args.cluster_mode, cluster_action=args.cluster_action,
        pt_interval=args.pt_interval, pt_schedule=args.pt_schedule,
        overlap_cluster_update_interval=args.overlap_cluster_update_interval,
        overlap_cluster_build_mode=args.overlap_cluster_build_mode,
        overlap_cluster_mode=args.overlap_cluster_mode,
        overlap_cluster_action=args.overlap_cluster_action,
        collect_cluster_stats=args.collect_cluster_stats,
        autocorrelation_max_lag=args.autocorrelation_max_lag,
        autocorrelation_backend=args.autocorrelation_backend,
        equilibration_diagnostic=args.equilibration_diagnostic,
    )


def print_table(model, out=None):
    """Per-temperature table of `simulate` (cli.py:656-689): T, E, Binder, C_v and, when present,
    the overlap Binder ratio, the mean FK cluster size and the top-4 overlap clusters."""
    out = out or sys.stdout
    columns = [("T", 8, lambda i: f"{model.temperatures[i]:8.4f}"),
               ("E", 10, lambda i: f"{model.energies_avg[i]:10.6f}"),
               ("Binder", 10, lambda i: f"{model.binder_cumulant[i]:10.6f}"),
               ("C_v", 10, lambda i: f"{model.heat_capacity[i]:10.4f}")]
    if hasattr(model, "sg_binder"):
        columns.append(("Overlap Binder", 15, lambda i: f"{model.sg_binder[i]:15.6f}"))
    if hasattr(model, "mean_cluster_size"):
        columns.append(("Cluster Size", 14, lambda i: f"{model.mean_cluster_size[i]:14.2f}"))
    if hasattr(model, "top_cluster_sizes"):
        def top4(i):
            a, b, c, d = model.top_cluster_sizes[i][:4]
            return f"({a:.3f}, {b:.3f}, {c:.3f}, {d:.3f})".rjust(30)
        columns.append(("Top-4 Clusters", 30, top4))
    header = "  ".join(f"{name:>{width}}" for name, width, _ in columns)
    print(header, file=out)
    print("-" * len(header), file=out)
    for i in range(len(model.temperatures)):
        print("  ".join(cell(i) for _, _, cell in columns), file=out)


def run_simulate(args):
    model = build_model(args)
    result = model.sample(args.n_sweeps, **sample_kwargs(args), warmup_ratio=args.warmup_ratio)
    print_table(model)
    if args.output:
        np.savez(args.output, **sweep_mod.simulate_arrays(model, result))
        print(f"\nResults saved to {args.output}")
    return model


def run_bench(args):
    model = build_model(args)
    t0 = time.perf_counter()
    model.sample(args.n_sweeps, **sample_kwargs(args), warmup_ratio=0.0)
    elapsed = time.perf_counter() - t0
    shape = "x".join(str(s) for s in args.shape)
    print(f"Lattice: {shape}  |  Temps: {args.n_temps}  |  Sweeps: {args.n_sweeps}")
    print(f"Total: {elapsed:.3f} s  |  {elapsed / args.n_sweeps * 1000:.3f} ms/sweep")
    return elapsed


def load_sweep_config(path):
    """TOML file -> sweep keywords (sections lattice / temperatures / replicas / sampling / cluster /
    parallel_tempering / overlap_cluster / diagnostics / output)."""
    with open(path, "rb") as f:
        cfg = tomllib.load(f)
    kw = {}
    for section, key, name, convert in TOML_MAP:
        if key in cfg.get(section, {}):
            value = cfg[section][key]
            kw[name] = convert(value) if convert else value
    autocorr = cfg.get("diagnostics", {}).get("autocorrelation", {})
    for key, name in TOML_AUTOCORR:
        if key in autocorr:
            kw[name] = autocorr[key]
    return kw


def sweep_settings(args):
    """Merge TOML file < command-line flags < SWEEP_DEFAULTS for what is still unset."""
    kw = load_sweep_config(args.config) if args.config is not None else {}
    for name in SWEEP_DEFAULTS:
        value = getattr(args, name, None)
        if value is not None:
            kw[name] = value
    for name, default in SWEEP_DEFAULTS.items():
        kw.setdefault(name, default)
    for name, flags in (("sizes", "--sizes"), ("temp_min", "--temp-min and --temp-max"),
                        ("temp_max", "--temp-min and --temp-max"), ("n_sweeps", "--n-sweeps")):
        if kw[name] is None:
            verb = "are" if " and " in flags else "is"
            print(f"error: {flags} {verb} required (via CLI or config file)", file=sys.stderr)
            sys.exit(1)
    if isinstance(kw["sizes"][0], str):
        kw["sizes"] = [tuple(int(x) for x in s.split(",")) for s in kw["sizes"]]
    if isinstance(kw["neighbor_offsets"], str):
        kw["neighbor_offsets"] = json.loads(kw["neighbor_offsets"])
    return kw


def run_sweep_cli(args):
    kw = sweep_settings(args)
    return sweep_mod.run_sweep(
        kw["sizes"], couplings=tuple(kw["couplings"]),
        temperatures=temperature_grid(kw["temp_min"], kw["temp_max"], kw["n_temps"],
                                      kw["temp_scale"]),
        n_replicas=kw["n_replicas"], n_disorder=kw["n_disorder"], seed=kw["seed"],
        neighbor_offsets=kw["neighbor_offsets"], geometry=kw["geometry"], n_sweeps=kw["n_sweeps"],
        sweep_mode=kw["sweep_mode"], cluster_update_interval=kw["cluster_interval"],
        cluster_mode=kw["cluster_mode"], cluster_action=kw["cluster_action"],
        pt_interval=kw["pt_interval"], pt_schedule=kw["pt_schedule"],
        overlap_cluster_update_interval=kw["overlap_cluster_update_interval"],
        overlap_cluster_build_modes=tuple(kw["overlap_cluster_build_mode"]),
        overlap_cluster_modes=tuple(kw["overlap_cluster_mode"]),
        overlap_cluster_action=kw["overlap_cluster_action"], warmup_ratio=kw["warmup_ratio"],
        collect_cluster_stats=kw["collect_cluster_stats"],
        autocorrelation_max_lag=kw["autocorrelation_max_lag"],
        autocorrelation_backend=kw["autocorrelation_backend"],
        autocorrelation_plot_temp=kw["autocorrelation_plot_temp"],
        equilibration_diagnostic=kw["equilibration_diagnostic"], save_plots=kw["save_plots"],
        save_data=kw["save_data"], output_dir=kw["output_dir"], sequential=kw["sequential"],
        snapshot_interval=kw["snapshot_interval"],
        device=args.device if args.device is not None else 0)


def build_parser():
    parser = argparse.ArgumentParser(
        prog="peapods_amd", description="Ising Monte Carlo simulations on MI355X from the command line.")
    sub = parser.add_subparsers(dest="command")
    simulate = sub.add_parser("simulate", help="Run an Ising simulation")
    _add_model_flags(simulate)
    simulate.add_argument("--warmup-ratio", type=float, default=0.25)
    simulate.add_argument("-o", "--output", type=str, default=None,
                          help="Save full results to .npz file")
    bench = sub.add_parser("bench", help="Benchmark sampling performance")
    _add_model_flags(bench)
    sweep = sub.add_parser("sweep", help="Run parameter sweeps")
    _add_sweep_flags(sweep)
    return parser


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    handlers = {"simulate": run_simulate, "bench": run_bench, "sweep": run_sweep_cli}
    if args.command not in handlers:
        parser.print_help()
        sys.exit(1)
    handlers[args.command](args)


if __name__ == "__main__":
    main()
