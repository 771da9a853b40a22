#!/usr/bin/env python3
"""``pangaea.py`` -- the reference orchestrator (/root/reference/src/pangaea.py) with steps 1-3 on the GPU.

Same command line (every flag of pangaea.py:130-171), same four resumable steps and marker files
(``1.features/feature_finished``, ``2.vae/model_finished``, ``3.clustering/clustering_finished``; pangaea.py:23-35),
same output tree.  Step 4 (sub-assembly + ensemble; clustering.py:132-164) belongs to the unchanged reassembly stage:
it is delegated to the reference's own ``final_assemble`` when a Pangaea checkout is given with ``--reference_src``
(or PANGAEA_SRC), and skipped with a log line otherwise.

    python -m pangaea_amd.pangaea -i interleaved.sorted.fastq -o out -c 30 -st 1,2,3
Multi-GPU: ``torchrun --nproc-per-node N -m pangaea_amd.pangaea ...`` shards step 1 (see pangaea_amd/dist.py).
"""
from __future__ import annotations

import argparse
import logging
import os
import sys

import numpy as np
import torch


def check_steps_finish(args, step):
    marks = {"1": ("1.features", "feature_finished"), "2": ("2.vae", "model_finished"),
             "3": ("3.clustering", "clustering_finished"), "4": ("4.assembly", "assemble_finished")}
    if step not in marks:
        return False
    d, f = marks[step]
    return os.path.exists(os.path.join(args.output, d, f))


def check_steps_required(args_steps, step):
    return step in ("1", "2", "3", "4") and step in args_steps


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument("-1", "--reads1", default="", help="path to reads1 file (linked-reads)")
    p.add_argument("-2", "--reads2", default="", help="path to reads2 file (linked-reads)")
    p.add_argument("-i", "--interleaved_reads", default="", help="path to reads file (long-reads)")
    p.add_argument("-o", "--output", required=True, help="output directory")
    p.add_argument("-l", "--min_length", type=int, default=2000, help="min barcode length (default 2000)")
    p.add_argument("-k", "--kmer", type=int, default=15, help="kmer for abundance (default 15)")
    p.add_argument("-tnf_k", "--tnf_kmer", type=int, default=4, help="kmer for TNF (default 4, long reads should use 3)")
    p.add_argument("-s", "--window_size", type=int, default=10, help="window size for abundance (default 10)")
    p.add_argument("-v", "--vector_size", type=int, default=400, help="vector size for abundance (default 400)")
    p.add_argument("-r", "--lr", type=float, default=0.005, help="learning rate (default 0.005)")
    p.add_argument("-w", "--weight_decay", type=float, default=0.0001, help="weight decay (default 0.0001)")
    p.add_argument("-e", "--epochs", type=int, default=100, help="number of epochs (default 100)")
    p.add_argument("-b", "--batch_size", type=int, default=2048, help="batch size (defult 2048)")
    p.add_argument("-d", "--dropout", type=float, default=0.2, help="dropout (default 0.2)")
    p.add_argument("-p", "--patience", type=int, default=20, help="early stop patience (default 20)")
    p.add_argument("-wa", "--weight_alpha", type=float, default=0.1, help="training weight for abundance and tnf (default 0.1)")
    p.add_argument("-wk", "--weight_kl", type=float, default=0.015, help="training weight for KL (default 0.015)")
    p.add_argument("-ld", "--latent_dim", type=int, default=32, help="latent dimension (default 32)")
    p.add_argument("-c", "--clusters", type=int, required=False, help="number of clusters")
    p.add_argument("-m", "--metaphlan_db", type=str, default="metaphlan_db", help="path to metaphlan db (default metaphlan_db)")
    p.add_argument("-t", "--threads", type=int, default=100, help="number of threads (default 100)")
    p.add_argument("-g", "--use_cuda", type=bool, default=True, help="use the GPU (default True; the feature path has no CPU form)")
    p.add_argument("-n", "--num_gpus", type=int, default=1, help="use gpu in parallel (if use cuda)")
    p.add_argument("-sp", "--spades", type=str, help="path to original contigs")
    p.add_argument("-lc", "--local_assembly", type=str, help="path to local assembly contigs")
    p.add_argument("-at", "--athena", type=str, help="path to athena contigs")
    p.add_argument("-lt", "--low_abd_cut", type=str, default="10,30", help="coverage for low abundance contigs")
    p.add_argument("-la", "--low_assembler", type=str, default="megahit", help="local assembly method (spades or megahit)")
    p.add_argument("-md", "--model", type=str, default="vae", help="model ( vae)")
    p.add_argument("-ls", "--loss_type", type=str, default="ce", help="reconstruction loss type (default ce)")
    p.add_argument("-st", "--steps", type=str, default="1,2,3,4", help="steps to run (default 1:feature extraction, 2:vae trainning, 3:clutsering, 4:sub-assembly and final assembly)")
    # additive option (not in the reference)
    p.add_argument("--reference_src", type=str, default=os.environ.get("PANGAEA_SRC", ""),
                   help="src/ directory of a Pangaea checkout: where step 4 and the metaphlan helper scripts live")
    return p


def _encode_sharded(args, feat, vae, model_path):
    """step 2's encode on several ranks: the rows never left the rank that made them, so every rank normalises
    and encodes its own block with a replica of the trained network (2.2 MB of weights travel, not gigabytes of count
    matrices), and only ``mu`` [N, 32] and the names are gathered on rank 0, which writes the reference's files.
    Returns (this rank's mu, all ranks' mu on rank 0 / None elsewhere).  The weights come over the control plane: the other
    ranks wait here for as long as rank 0 trains."""
    import torch.distributed as dist
    from . import dist as pdist
    from .data import Data
    from .models.VAENET import VAENET
    rank = dist.get_rank()
    names, tnf, abd = feat.local
    state = None
    if rank == 0:
        state = {k: v.cpu() for k, v in torch.load(os.path.join(model_path, "train_model.pk"), map_location="cpu").items()}
    else:
        vae = VAENET(abd_dim=abd.shape[1], tnf_dim=tnf.shape[1], latent_size=args.latent_dim, num_classes=args.clusters,
                     epochs=args.epochs, cuda=torch.cuda.is_available(), num_gpus=args.num_gpus, lr=args.lr, dropout=args.dropout,
                     alpha=args.weight_alpha, w_kl=args.weight_kl, weight_decay=args.weight_decay)
    state = pdist.agreed(state)
    vae.network.load_state_dict(state)
    vae.network.eval()
    mu = vae.encode(Data(np.asarray(names, dtype=object), abd, tnf))
    all_mu = pdist.gather_rows(mu, dst=0)
    gathered = [None] * dist.get_world_size() if rank == 0 else None
    dist.gather_object(list(names), gathered, dst=0, group=pdist.control_group())
    if rank == 0:
        VAENET.write_latent(model_path, all_mu.cpu().numpy(), [n for part in gathered for n in part])
    return mu, all_mu


def run(args, script_path):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    from . import dist as pdist
    if world > 1 and not torch.distributed.is_initialized():
        import datetime
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
        # (PANGAEA_DIST_TIMEOUT_S: the data group's collective timeout -- RCCL's default is 10 minutes; tests shorten it to show
        # that no rank waits for another rank's host work inside a data collective)
        timeout = os.environ.get("PANGAEA_DIST_TIMEOUT_S")
        kw = {"timeout": datetime.timedelta(seconds=float(timeout))} if timeout else {}
        torch.distributed.init_process_group(os.environ.get("PANGAEA_DIST_BACKEND", "nccl"), **kw)
    pdist.control_group()               # (every rank, here: waits for rank 0's host-side work go over this group)
    try:
        _run(args, script_path)
    except BaseException:
        # a rank that fails leaves at once (no barrier: its peers may be anywhere; the launcher ends them)
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        raise
    finally:
        pdist.leave()                   # (a no-op when _run has left already, as it does behind its last collective)


def _run(args, script_path):
    from . import dist as pdist
    from .clustering import cluster_barcode_reads
    from .data import Data
    from .feature import Feature
    from .loader import shuffled_batches, weighted_batches
    from .models.VAENET import VAENET
    from .utils import init_all

    multi = pdist.is_distributed()
    rank0 = not multi or torch.distributed.get_rank() == 0
    init_all(seed=2021, threads=min(args.threads, os.cpu_count() or 1), logfile="log", level=logging.INFO, outdir=args.output,
             file_log=rank0)
    logging.info("command: " + " ".join(sys.argv))
    logging.info(args)

    model_path = os.path.join(args.output, "2.vae")
    cluster_path = os.path.join(args.output, "3.clustering")
    assembly_path = os.path.join(args.output, "4.assembly")
    have_reads = (args.reads1 and args.reads2) or args.interleaved_reads
    read_specify = abundance = tnf = None
    feat = None

    agreed = pdist.agreed               # rank 0's reading of the output directory decides for everybody (control plane)

    # step 1: feature extraction (every rank takes part: its own runs, one table exchange; rank 0 writes the caches)
    if not check_steps_required(args.steps, "1"):
        logging.info("skip step 1: feature extraction")
    elif agreed(check_steps_finish(args, "1")):
        logging.info("step 1: feature extraction finished")
    elif have_reads:
        feat = Feature(args, script_path)
        read_specify, abundance, tnf = feat.extract_features()
    else:
        print("Please provide one or two input file(s):-1 and -2 for pair-end linked reads; -lr as long reads; -i for interleaved linked reads.")
        sys.exit()

    # step 2: training (rank 0: the loop is small and sequential) + encode (every rank, its own rows)
    mu_local = mu_all = None
    if not check_steps_required(args.steps, "2"):
        logging.info("skip step 2: training")
    elif agreed(check_steps_finish(args, "2")):
        logging.info("step 2: training finished")
    else:
        # rows still sharded in memory on every rank -> replicated encode; otherwise (resume from the cache files) rank 0 alone
        sharded = multi and pdist.everyone(feat is not None and feat.local is not None)
        vae = None
        if rank0:
            if not all(isinstance(a, np.ndarray) for a in (read_specify, abundance, tnf)):
                read_specify, abundance, tnf = Feature(args, script_path).load_features()
            dataset = Data(read_specify, abundance, tnf)
            test_size = min(int(len(dataset) * 0.7), 1000000)
            train = weighted_batches(dataset, args.batch_size)
            test = weighted_batches(dataset, args.batch_size, num_samples=test_size, replacement=False)
            original = shuffled_batches(dataset, args.batch_size)
            os.makedirs(model_path, exist_ok=True)
            vae = VAENET(abd_dim=abundance.shape[1], tnf_dim=tnf.shape[1], latent_size=args.latent_dim, num_classes=args.clusters,
                         epochs=args.epochs, cuda=True, num_gpus=args.num_gpus, lr=args.lr, dropout=args.dropout,
                         alpha=args.weight_alpha, w_kl=args.weight_kl, weight_decay=args.weight_decay)
            vae.train(train, test, original, model_path, args.patience, encode=not sharded)
        if sharded:
            mu_local, mu_all = _encode_sharded(args, feat, vae, model_path)

    # step 3, its collective part: with the latent rows still sharded, the Lloyd iterations of RPH-KMeans run on every rank's
    # own rows (an all-reduce of the [k, 32] sums and [k] counts per iteration); rank 0 reduces the points and seeds
    labels = None
    do3 = check_steps_required(args.steps, "3") and not agreed(check_steps_finish(args, "3"))
    if multi and do3 and pdist.everyone(mu_local is not None and bool(args.clusters)
                                         and (not rank0 or not os.path.isfile(os.path.join(cluster_path, "clusters.tsv")))):
        from .clustering import clustering_rph_kmeans_sharded
        logging.info("start clustering (Lloyd iterations on every rank's rows)")
        labels = clustering_rph_kmeans_sharded(mu_local, mu_all, args.clusters)
    # the last collective is behind us: every rank leaves the group, rank 0 goes on alone (bin extraction and the assembly
    # stage take minutes to hours of host time -- nobody may wait in a collective for them)
    pdist.leave()
    if not rank0:
        return

    # step 3: clustering
    if not check_steps_required(args.steps, "3"):
        logging.info("skip step 3: clustering")
    elif check_steps_finish(args, "3"):
        logging.info("step 3: clustering finished")
    else:
        logging.info("start clustering")
        os.makedirs(cluster_path, exist_ok=True)
        if have_reads:
            cluster_barcode_reads(args, model_path, cluster_path, args.reference_src or script_path, clusters=labels)
        else:
            logging.info("Please provide one or two input file(s):-1 and -2 for pair-end linked reads; -lr as long reads; -i for interleaved linked reads.")
            sys.exit()

    # step 4: the unchanged reassembly / ensemble stage of the reference
    if not check_steps_required(args.steps, "4"):
        logging.info("skip step 4: assembly")
    elif check_steps_finish(args, "4"):
        logging.info("step 4: assembly finished")
    elif args.reference_src and os.path.isfile(os.path.join(args.reference_src, "clustering.py")):
        logging.info("start assembly (reference final_assemble)")
        sys.path.insert(0, args.reference_src)
        sys.path.insert(0, os.path.join(os.path.dirname(args.reference_src), "third_parties", "rph_kmeans"))
        from clustering import final_assemble              # the reference's own step 4
        final_assemble(args, cluster_path, assembly_path, args.reference_src)
    else:
        logging.info("step 4 (multi-threshold reassembly + ensemble) is the reference's unchanged stage: "
                     "give --reference_src <Pangaea>/src to run it on " + cluster_path)
    logging.info("program finished successfully")


def main(argv=None):
    args = build_parser().parse_args(argv)
    run(args, os.path.dirname(os.path.abspath(__file__)))


if __name__ == "__main__":
    main()
