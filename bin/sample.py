#!/usr/bin/env python3
"""
Sample protein backbones from a trained foldingdiff model on MI355X -- stand-in for the output stage of the
reference's ``bin/sample.py`` (same flags, same files):

    <outdir>/model_snapshot/                      copy of the model directory (from_dir(copy_to=...), :341-343)
    <outdir>/sampled_angles/generated_{i}.csv.gz  final angles of every sampled backbone (:360-368)
    <outdir>/sampled_pdb/generated_{i}.pdb        N-CA-C backbones built by NeRF on the device (:369, :105-128)
    with --fullhistory: sampled_angles/sample_history/generated_{i}/generated_{i}_timestep_{t}.csv.gz and
                        sampled_pdb/sample_history/generated_{i}/generated_{i}_timestep_{t}.pdb (:371-398)

What is NOT here (outside the sampler path): the matplotlib / astropy plots, secondary-structure
annotation and the --testcomparison statistics against the CATH test set (they need the dataset pipeline);
--testcomparison therefore raises, --nopsea is accepted and ignored.  The model must be a local directory
(no network): training_args.json, config.json, models/best_by_valid/*.ckpt [, training_mean_offset.npy].

One process per GPU under ``torchrun`` shards every batch over the GPUs (sampling.sample); rank 0 writes the files.
"""
import argparse
import json
import logging
import os
import sys
from pathlib import Path

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import pandas as pd  # noqa: E402
import torch  # noqa: E402

from foldingdiff_amd import modelling, sampling  # noqa: E402
from foldingdiff_amd.angles_and_coords import write_preds_pdb_folder  # noqa: E402
from foldingdiff_amd.datasets import AnglesEmptyDataset, NoisedAnglesDataset  # noqa: E402

# the value the reference's default seed expression (bin/sample.py:34-37) evaluates to
SEED = 7344


def build_datasets(model_dir: Path) -> NoisedAnglesDataset:
    """The data-free dataset shell of the reference's ``build_datasets(load_actual=False)`` (bin/sample.py:49-93)."""
    with open(model_dir / "training_args.json") as source:
        training_args = json.load(source)
    dset = AnglesEmptyDataset.from_dir(str(model_dir))
    return NoisedAnglesDataset(
        dset=dset,
        dset_key="coords" if training_args["angles_definitions"] == "cart-coords" else "angles",
        timesteps=training_args["timesteps"],
        exhaustive_t=False,
        beta_schedule=training_args["variance_schedule"],
        nonangular_variance=1.0,
        angular_variance=training_args["variance_scale"],
    )


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("-m", "--model", type=str, required=True,
                        help="Path to model directory: training_args.json, config.json and a models folder at a minimum")
    parser.add_argument("--outdir", "-o", type=str, default=os.getcwd(), help="Path to output directory")
    parser.add_argument("--num", "-n", type=int, default=10, help="Number of examples to generate *per length*")
    parser.add_argument("-l", "--lengths", type=int, nargs=2, default=[50, 128], help="Range of lengths to sample from")
    parser.add_argument("-b", "--batchsize", type=int, default=512, help="Batch size to use when sampling")
    parser.add_argument("--fullhistory", action="store_true", help="Store full history, not just final structure")
    parser.add_argument("--testcomparison", action="store_true", help="(not available: needs the CATH data pipeline)")
    parser.add_argument("--nopsea", action="store_true", help="accepted for compatibility; no PSEA step exists here")
    parser.add_argument("--seed", type=int, default=SEED, help="Random seed")
    parser.add_argument("--device", type=str, default="cuda:0", help="Device to use")
    return parser


def main(argv=None) -> None:
    args = build_parser().parse_args(argv)
    if args.testcomparison:
        raise NotImplementedError("--testcomparison needs the CATH test split and the plotting stack, which are outside this path")
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    outdir = Path(args.outdir)
    # EVERY rank evaluates the same preconditions, before anything is created and before the process group exists; the verdicts
    # are then all-reduced so that either all ranks raise or none does (a rank that exits alone leaves the others blocked in
    # init_process_group / the first collective until the rendezvous or RCCL timeout)
    problem = ""
    if not os.path.isdir(args.model):
        problem = f"{args.model} is not a local model directory (there is no network access here)"
    elif os.path.isdir(outdir) and os.listdir(outdir):
        problem = f"Expected {outdir} to be empty!"  # Be extra cautious so we don't overwrite any results (bin/sample.py:299)
    if world > 1:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        args.device = f"cuda:{local_rank}"
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(args.device))
        from foldingdiff_amd import distributed as fdist
        if fdist.any_rank_failed(bool(problem), device=torch.device(args.device)):
            dist.destroy_process_group()
            raise AssertionError(problem or "another rank failed its start-up checks (see its log)")
        if sampling.NOISE_MODE == "torch":
            # the reference's draw order needs every rank to draw the WHOLE batch's stream on one host thread: it reproduces the
            # single-process samples bit for bit, but it does not scale (12.6 GB of torch.randn per rank per batch of 4096 x 128)
            logging.warning("world size %d with the default noise mode 'torch': every rank draws the whole batch's noise on the host "
                            "(bit-identical to a single-process run, but it will not scale); set FOLDINGDIFF_AMD_NOISE=philox "
                            "for on-device noise keyed by the global sequence index", world)
    elif problem:
        raise AssertionError(problem)
    if rank == 0:
        os.makedirs(outdir, exist_ok=True)

    train_dset = build_datasets(Path(args.model))
    model = modelling.BertForDiffusionBase.from_dir(
        args.model, copy_to=str(outdir / "model_snapshot") if rank == 0 else "").to(torch.device(args.device))

    sweep_min_len, sweep_max_len = args.lengths
    assert sweep_min_len < sweep_max_len
    assert sweep_max_len <= train_dset.dset.pad

    torch.manual_seed(args.seed)
    sampled = sampling.sample(model, train_dset, n=args.num, sweep_lengths=(sweep_min_len, sweep_max_len),
                              batch_size=args.batchsize, final_only=not args.fullhistory)
    if rank == 0:
        names = train_dset.feature_names["angles"]
        sampled_dfs = [pd.DataFrame(s[-1], columns=names) for s in sampled]
        sampled_angles_folder = outdir / "sampled_angles"
        os.makedirs(sampled_angles_folder, exist_ok=True)
        logging.info(f"Writing sampled angles to {sampled_angles_folder}")
        for i, s in enumerate(sampled_dfs):
            s.to_csv(sampled_angles_folder / f"generated_{i}.csv.gz")
        write_preds_pdb_folder(sampled_dfs, str(outdir / "sampled_pdb"))
        if args.fullhistory:
            full_history_angles_dir = sampled_angles_folder / "sample_history"
            os.makedirs(full_history_angles_dir)
            full_history_pdb_dir = outdir / "sampled_pdb" / "sample_history"
            os.makedirs(full_history_pdb_dir)
            for i, sampled_series in enumerate(sampled):
                snapshot_dfs = [pd.DataFrame(snapshot, columns=names) for snapshot in sampled_series]
                ith_angle_dir = full_history_angles_dir / f"generated_{i}"
                os.makedirs(ith_angle_dir, exist_ok=True)
                for timestep, snapshot_df in enumerate(snapshot_dfs):
                    snapshot_df.to_csv(ith_angle_dir / f"generated_{i}_timestep_{timestep}.csv.gz")
                write_preds_pdb_folder(snapshot_dfs, str(full_history_pdb_dir / f"generated_{i}"),
                                       basename_prefix=f"generated_{i}_timestep_")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    main()
