"""world_size-2 checks of the multi-GPU decomposition over gloo: run sharding, the variable-length
all-gather used for hash tables, and the dense-table all-reduce.  On the CPU the kernels cannot run, so the oracle
plays the counter; the gpu-marked variant runs the real kernels on two ranks sharing cuda:0."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle
from pangaea_amd import dist as pdist
from pangaea_amd import kmer, synth
from pangaea_amd.reads import ReadStream

from .conftest import GOLDEN


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(fn, world, *args):
    port = _free_port()
    mp.spawn(fn, args=(world, port) + args, nprocs=world, join=True)


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def test_shards_tile_the_runs_and_the_characters():
    cfg = synth.SynthConfig(n_pairs=1500, n_barcodes=23, n_genomes=2, genome_len=30_000, fragment=8_000, n_rate=0.1)
    host = synth.generate(cfg)
    for world in (1, 2, 3, 8):
        names, total_valid = [], 0
        full_rows = host.rows(2000)
        k = 11
        want = oracle.Table(k).count(host.decode())
        got = {}
        for r in range(world):
            sh = pdist.shard_stream(host, r, world)
            assert sh.n_words % 256 == 0
            rows = sh.rows(2000)
            names += rows.names
            for a, b, nm in zip(rows.start, rows.end, rows.names):
                i = full_rows.names.index(nm)
                assert sh.decode(int(a), int(b)) == host.decode(int(full_rows.start[i]), int(full_rows.end[i]))
            c, n = oracle.Table(k).count(sh.decode()).items()
            for ci, ni in zip(c, n):
                got[int(ci)] = got.get(int(ci), 0) + int(ni)
        assert names == full_rows.names
        wc, wn = want.items()
        assert got == {int(c): int(n) for c, n in zip(wc, wn)}        # every character counted on exactly one rank


def _gloo_worker(rank, world, port, fastq):
    _init(rank, world, port)
    try:
        host = ReadStream.from_fastq(fastq)
        sh = pdist.shard_stream(host, rank, world)
        # hash-table form: compact (code << 22 | count) vectors of different lengths, gathered on every rank
        k = 21
        c, n = oracle.Table(k).count(sh.decode()).items()
        mine = torch.from_numpy(((c << np.uint64(22)) | n).astype(np.int64))
        parts = pdist.gather_pairs(mine)
        assert len(parts) == world and torch.equal(parts[rank], mine)
        merged = {}
        for p in parts:
            for v in p.numpy().view(np.uint64):
                merged[int(v >> np.uint64(22))] = merged.get(int(v >> np.uint64(22)), 0) + int(v & np.uint64((1 << 22) - 1))
        wc, wn = oracle.Table(k).count(host.decode()).items()
        assert merged == {int(a): int(b) for a, b in zip(wc, wn)}
        # dense form: the all-reduce of the path
        k = 6
        t = kmer.KmerTable.alloc(k, "cpu", "dense")
        c, n = oracle.Table(k).count(sh.decode()).items()
        t.data[torch.from_numpy(c.astype(np.int64))] = torch.from_numpy(n.astype(np.int32))
        pdist.exchange_table(t)
        wc, wn = oracle.Table(k).count(host.decode()).items()
        gc, gn = t.items()
        assert np.array_equal(gc, wc) and np.array_equal(gn, wn)
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo_cpu():
    _spawn(_gloo_worker, 2, os.path.join(GOLDEN, "tenx_mixed.fq"))


def _ingest_worker(rank, world, port, fastq, gz):
    _init(rank, world, port)
    try:
        whole = ReadStream.from_fastq(fastq)
        want = [(n, whole.decode(int(a), int(b))) for n, a, b in zip(whole.run_names, whole.run_off[:-1], whole.run_off[1:])]
        for path in (fastq, gz):            # cut by bytes at ingest / parsed whole and cut by runs
            sh = pdist.ingest_shard(path)
            mine = [(n, sh.decode(int(a), int(b))) for n, a, b in zip(sh.run_names, sh.run_off[:-1], sh.run_off[1:])]
            parts = [None] * world
            dist.all_gather_object(parts, mine)
            assert [r for p in parts for r in p] == want
            pairs = torch.tensor([sh.n_pairs])
            dist.all_reduce(pairs)
            assert path == gz or int(pairs.item()) == whole.n_pairs
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_ingest_their_own_byte_ranges(world, tmp_path):
    import gzip
    from pangaea_amd import synth
    cfg = synth.SynthConfig(n_pairs=3000, n_barcodes=23, n_genomes=2, genome_len=20_000, fragment=4_000, unbarcoded=0.1)
    fq, gz = str(tmp_path / "a.fq"), str(tmp_path / "a.fq.gz")
    synth.write_fastq(synth.generate(cfg), cfg, fq)
    with gzip.open(gz, "wb") as f:
        f.write(open(fq, "rb").read())
    _spawn(_ingest_worker, world, fq, gz)


def _gpu_worker(rank, world, port, fastq, outdir):
    _init(rank, world, port)
    try:
        from pangaea_amd import feature
        torch.cuda.set_device(0)
        names, tnf, abd = feature.compute_features(fastq, None, 21, 4, 1, 6, 100)
        if rank == 0:
            np.savez(os.path.join(outdir, "r.npz"), names=np.array(names), tnf=tnf, abd=abd)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_real_kernels_match_single_process(tmp_path):
    """two processes share cuda:0, shard the runs, exchange the hash table (gather + merge kernel) and gather rows"""
    fq = os.path.join(GOLDEN, "tenx_mixed.fq")
    _spawn(_gpu_worker, 2, fq, str(tmp_path))
    got = np.load(str(tmp_path / "r.npz"))
    rd = oracle.Reads(fq)
    table = oracle.Table(21).count(rd.all_seq())
    names, tnf, abd = rd.features(100, k_tnf=4, k_abd=21, table=table, window=1, vsize=6)
    assert list(got["names"]) == names
    assert np.array_equal(got["tnf"], tnf) and np.array_equal(got["abd"], abd)


def _rccl_worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        # the collectives of the exchange as RCCL sees them (a one-rank group is all a one-GPU box can hold)
        mine = torch.arange(1000, dtype=torch.int64, device="cuda") * 3
        out = torch.empty(1000, dtype=torch.int64, device="cuda")
        pdist._all_gather_flat(out, mine)
        assert torch.equal(out, mine)
        parts = pdist.gather_pairs(mine)
        assert len(parts) == 1 and torch.equal(parts[0], mine)
        t = torch.ones(8, dtype=torch.int32, device="cuda")
        dist.all_reduce(t)
        assert int(t.sum()) == 8
        work = pdist._all_gather_flat(out.zero_(), mine, async_op=True)
        work.wait()
        assert torch.equal(out, mine)
        # the whole bucketed exchange over RCCL (fills, range-wise asynchronous gathers, LDS rebuilds), deferred and direct
        cfg = synth.SynthConfig(n_pairs=3000, n_barcodes=24, n_genomes=3, genome_len=20_000, fragment=8_000, seed=51)
        s = synth.generate(cfg, device="cuda:0")
        want = kmer.KmerTable.with_slots(21, "cuda:0", 20, 0).count(s).items()
        for deferred, log2_slots in ((False, 20), (True, 20), (True, 21)):       # 2^21 slots: 2048 buckets, the 6-byte format
            table = kmer.KmerTable.with_slots(21, "cuda:0", log2_slots, 10)
            table.count(s, deferred_group=1) if deferred else table.count(s)
            pdist._exchange_bucketed(table)
            table.check_status()
            assert not table.pending and all(np.array_equal(x, y) for x, y in zip(table.items(), want))
        # the owner-partitioned form over RCCL (all-to-all + all-gather), as far as one rank can show it
        pdist.OWNER_MIN_WORLD = 1
        table = kmer.KmerTable.with_slots(21, "cuda:0", 21, 10)
        table.count(s, deferred_group=1)
        pdist._exchange_bucketed(table)
        table.check_status()
        assert not table.pending and all(np.array_equal(x, y) for x, y in zip(table.items(), want))
        pdist.OWNER_MIN_WORLD = 4
        # counts beyond 0xffff travel as 0xffff + a remainder in the overflow list: 3.1 M copies of one 21-mer
        s2 = ReadStream.from_runs([("a", b"A" * 1_600_000 + b"N" + b"T" * 1_500_040 + b"N"), ("b", b"ACGT" * 600 + b"N")], device="cuda:0")
        want2 = kmer.KmerTable.with_slots(21, "cuda:0", 21, 10).count(s2).items()
        table = kmer.KmerTable.with_slots(21, "cuda:0", 21, 10)
        table.count(s2, deferred_group=0)
        pdist._exchange_bucketed(table)
        table.check_status()
        got2 = table.items()
        assert np.array_equal(got2[0], want2[0]) and np.array_equal(got2[1], want2[1]) and got2[1].max() >= 1 << 21
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_collectives_of_the_exchange_run():
    _spawn(_rccl_worker, 1)


def _exchange_worker(rank, world, port, outdir, deferred):
    _init(rank, world, port)
    try:
        torch.cuda.set_device(0)
        if world == 3:
            pdist.OWNER_MIN_WORLD = 3          # uneven owner ranges (2048 buckets over 3 owners)
        cfg = synth.SynthConfig(n_pairs=3000, n_barcodes=24, n_genomes=3, genome_len=20_000, fragment=8_000, seed=51)
        s = synth.generate(cfg, device="cuda:0")
        bounds = [0] + [s.n_words * (r + 1) // world + 7 * (r + 1) for r in range(world - 1)] + [s.n_words]
        w0, w1 = bounds[rank], bounds[rank + 1]
        # 1024 / 2048 buckets: exchanged in four ranges; from 2^11 buckets on a deferred count travels in the 6-byte format
        table = kmer.KmerTable.with_slots(21, "cuda:0", 21 if deferred == "planes" else 20, 10)
        assert table.n_buckets >= 64 * pdist.EXCHANGE_RANGES and (table.tag_bits <= 31) == (deferred == "planes")
        if deferred:
            table.count(s, w0, w1, deferred_group=1)
            assert table.pending
        else:
            table.count(s, w0, w1)
        pdist.exchange_table(table)
        assert not table.pending
        c, n = table.items()
        np.savez(os.path.join(outdir, f"t{rank}.npz"), c=c, n=n)
        if deferred == "planes":
            # counts beyond 0xffff (3.1 M copies of one 21-mer on EVERY rank): 0xffff in the planes + remainders in the overflow
            # lists, in both phases of the owner-partitioned form; the sum saturates exactly
            s2 = ReadStream.from_runs([("a", b"A" * 1_600_000 + b"N" + b"T" * 1_500_040 + b"N"), ("b", b"ACGT" * 600 + b"N")], device="cuda:0")
            want = kmer.KmerTable.with_slots(21, "cuda:0", 21, 10)
            for _ in range(world):
                want.count(s2)
            table = kmer.KmerTable.with_slots(21, "cuda:0", 21, 10)
            table.count(s2, deferred_group=0)
            pdist.exchange_table(table)
            got, exp = table.items(), want.items()
            assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]) and got[1].max() == 1 << 21
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("deferred", [False, True, "planes"])
def test_two_ranks_exchange_in_bucket_ranges(tmp_path, deferred):
    """each rank counts half of a stream (directly / in deferred form); after the exchange -- fills, compaction, four
    range-wise gathers and LDS rebuilds -- both hold the table of the whole stream"""
    _check_exchange(tmp_path, 2, deferred)


def _check_exchange(tmp_path, world, deferred):
    _spawn(_exchange_worker, world, str(tmp_path), deferred)
    cfg = synth.SynthConfig(n_pairs=3000, n_barcodes=24, n_genomes=3, genome_len=20_000, fragment=8_000, seed=51)
    s = synth.generate(cfg, device="cuda:0")
    want = kmer.KmerTable.with_slots(21, "cuda:0", 20, 0).count(s).items()
    for r in range(world):
        got = np.load(str(tmp_path / f"t{r}.npz"))
        assert np.array_equal(got["c"], want[0]) and np.array_equal(got["n"], want[1])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [4, 3])
def test_owner_partitioned_exchange(tmp_path, world):
    """from four ranks on the partial tables are reduced at bucket-range owners (all-to-all, LDS rebuild of the owned range)
    and the merged ranges are all-gathered; every rank ends with the table of the whole stream (3 ranks: uneven ranges)"""
    _check_exchange(tmp_path, world, "planes")


# ------------------------------------------------------------------ rows stay sharded; rank 0 writes; encode is replicated


def _rows_worker(rank, world, port, tmp):
    _init(rank, world, port)
    try:
        # blocks of different lengths and dtypes reach rank 0 as they are (no widening, no replication)
        for dtype, width in ((torch.int32, 7), (torch.float32, 32), (torch.int64, 1)):
            n = [5, 0, 9][rank % 3]
            mine = (torch.arange(n * width).reshape(n, width) + 1000 * rank).to(dtype)
            got = pdist.gather_rows(mine, dst=0)
            if rank == 0:
                want = torch.cat([(torch.arange([5, 0, 9][r % 3] * width).reshape(-1, width) + 1000 * r).to(dtype) for r in range(world)])
                assert got.dtype == dtype and torch.equal(got, want)
            else:
                assert got is None
        assert pdist.everyone(True) and not pdist.everyone(rank != 1)
        # the replicated encode of pangaea.run: every rank encodes its own rows with rank 0's weights, mu gathered on rank 0
        import argparse
        from pangaea_amd import pangaea
        from pangaea_amd.data import Data
        from pangaea_amd.models.VAENET import VAENET
        rs = np.random.RandomState(7)
        abd = rs.poisson(2.0, size=(37, 400)).astype(np.int32)
        tnf = rs.poisson(30.0, size=(37, 136)).astype(np.int32)
        names = [f"bc{i:03d}" for i in range(37)]
        cut = [0, 11, 37] if world == 2 else [0, 11, 20, 37]
        args = argparse.Namespace(latent_dim=32, clusters=5, epochs=1, num_gpus=1, lr=0.005, dropout=0.2, weight_alpha=0.1,
                                  weight_kl=0.015, weight_decay=0.0001)
        model = os.path.join(tmp, "2.vae")
        vae = None
        if rank == 0:
            os.makedirs(model, exist_ok=True)
            torch.manual_seed(5)
            vae = VAENET(400, 136, 32, 5, 1, False, 1, 0.005, 0.2, 0.1, 0.015, 0.0001)
            torch.save(vae.network.state_dict(), os.path.join(model, "train_model.pk"))
        dist.barrier()

        class Feat:
            local = (names[cut[rank]:cut[rank + 1]], tnf[cut[rank]:cut[rank + 1]], abd[cut[rank]:cut[rank + 1]])
        pangaea._encode_sharded(args, Feat, vae, model)
        dist.barrier()
        if rank == 0:
            latent = np.load(os.path.join(model, "latent.npz"))["arr_0"]
            bcs = np.load(os.path.join(model, "barcodes.npz"))["arr_0"]
            assert list(bcs) == names and os.path.isfile(os.path.join(model, "model_finished"))
            vae.network.eval()
            whole = vae.encode(Data(np.array(names, dtype=object), abd, tnf, device="cpu")).numpy()
            assert latent.shape == (37, 32) and np.abs(latent - whole).max() <= 1e-5 * np.abs(whole).max()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_rows_stay_sharded_and_the_encode_is_replicated(world, tmp_path):
    _spawn(_rows_worker, world, str(tmp_path))


def _pipeline_worker(rank, world, port, fq, out):
    # the data group's collectives time out after 8 s, and rank 0 is slowed down by 10 s in its training and again in its tail
    # (bin extraction): the other rank waits for the weights on the control plane and is gone before the tail starts
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      PANGAEA_DIST_BACKEND="gloo", PANGAEA_DIST_TIMEOUT_S="8")
    import time
    from pangaea_amd import clustering, pangaea
    from pangaea_amd.models.VAENET import VAENET
    train, tail = VAENET.train, clustering.cluster_barcode_reads
    if rank == 0:
        VAENET.train = lambda self, *a, **kw: (time.sleep(10), train(self, *a, **kw))[1]
        clustering.cluster_barcode_reads = lambda *a, **kw: (time.sleep(10), tail(*a, **kw))[1]
    pangaea.main(["-i", fq, "-o", out, "-c", "4", "-k", "21", "-l", "2000", "-e", "2", "-b", "32", "-st", "1,2,3", "-t", "2"])
    assert not dist.is_initialized()                    # every rank left the group behind


@pytest.mark.gpu
def test_two_rank_pipeline_on_a_shared_output_directory(tmp_path):
    """torchrun-style run of the orchestrator with two ranks (gloo, sharing cuda:0) into ONE output directory: the cache
    files are those of a one-rank run (rank 0 alone writes them, complete), the rows are the oracle's, latent.npz holds
    every barcode once in file order and equals the saved network's encode of the whole matrices"""
    import pandas as pd
    from pangaea_amd import pangaea
    from pangaea_amd.data import Data
    from pangaea_amd.models.VAENET import VAENET
    cfg = synth.SynthConfig(n_pairs=6000, n_barcodes=120, n_genomes=4, genome_len=100_000, fragment=20_000, seed=3)
    s = synth.generate(cfg)
    fq = str(tmp_path / "reads.sorted.fastq")
    synth.write_fastq(s, cfg, fq)
    out2, out1 = str(tmp_path / "two"), str(tmp_path / "one")
    mp.spawn(_pipeline_worker, args=(2, _free_port(), fq, out2), nprocs=2, join=True)
    pangaea.main(["-i", fq, "-o", out1, "-c", "4", "-k", "21", "-l", "2000", "-e", "2", "-b", "32", "-st", "1", "-t", "2"])
    for fn in ("tnf.m2000.gz", "abundance.k21.v400.w10.m2000.gz"):
        a = pd.read_csv(os.path.join(out1, "1.features", fn), header=None)
        b = pd.read_csv(os.path.join(out2, "1.features", fn), header=None)
        assert a.equals(b), fn
    assert not [f for f in os.listdir(os.path.join(out2, "1.features")) if ".tmp" in f]
    rd = oracle.Reads(fq)
    table = oracle.Table(21, threads=4).count(rd.all_seq())
    onames, otnf, oabd = rd.features(2000, k_tnf=4, k_abd=21, table=table, window=10, vsize=400, threads=4)
    abd = pd.read_pickle(os.path.join(out2, "1.features", "abundance.k21.v400.w10.m2000.pkl"))
    tnf = pd.read_pickle(os.path.join(out2, "1.features", "tnf.m2000.pkl"))
    assert list(abd[0]) == onames and np.array_equal(abd.drop(columns=0).to_numpy(), oabd) and np.array_equal(tnf.drop(columns=0).to_numpy(), otnf)
    latent = np.load(os.path.join(out2, "2.vae/latent.npz"))["arr_0"]
    bcs = np.load(os.path.join(out2, "2.vae/barcodes.npz"))["arr_0"]
    assert list(bcs) == onames and latent.shape == (120, 32)
    vae = VAENET(400, 136, 32, 4, 2, True, 1, 0.005, 0.2, 0.1, 0.015, 0.0001)
    vae.network.load_state_dict(torch.load(os.path.join(out2, "2.vae/train_model.pk"), map_location="cuda:0"))
    vae.network.eval()
    whole = vae.encode(Data(np.array(onames, dtype=object), oabd, otnf)).cpu().numpy()
    assert np.abs(latent - whole).max() <= 1e-5 * np.abs(whole).max()
    for rel in ("3.clustering/clusters.tsv", "3.clustering/clustering_finished", "2.vae/model_finished", "log"):
        assert os.path.isfile(os.path.join(out2, rel)), rel


# ------------------------------------------------------------------ sharded Lloyd (last bullet) and the control plane


def _blobs(n=3000, k=6, dim=32, seed=3):
    rs = np.random.RandomState(seed)
    centres = rs.normal(0, 6.0, size=(k, dim))
    which = rs.randint(0, k, size=n)
    return (centres[which] + rs.normal(0, 0.5, size=(n, dim))).astype(np.float32), which


def _lloyd_worker(rank, world, port):
    from pangaea_amd.clustering import RPHKMeans, clustering_rph_kmeans_sharded, lloyd
    _init(rank, world, port)
    try:
        x, _ = _blobs()
        cut = [0, 1100, 3000] if world == 2 else [0, 1100, 1100, 3000]              # (three ranks: one of them holds no row)
        mine = torch.from_numpy(x[cut[rank]:cut[rank + 1]])
        # (a) Lloyd from given centres, one of them so far out that its cluster starts empty: the relocation step too
        rs = np.random.RandomState(5)
        c0 = torch.from_numpy(np.concatenate([x[rs.choice(len(x), 5, replace=False)], np.full((1, 32), 1e3, dtype=np.float32)]))
        lab, cen, inertia, n_iter = lloyd(mine, c0, sharded=True)
        want_lab, want_cen, want_inertia, want_iter = lloyd(torch.from_numpy(x), c0)
        assert torch.equal(lab, want_lab[cut[rank]:cut[rank + 1]]) and n_iter == want_iter
        assert torch.allclose(cen, want_cen, rtol=1e-4, atol=1e-4) and abs(inertia - want_inertia) <= 1e-4 * want_inertia
        # (b) the whole RPH-KMeans call: rank 0 reduces and seeds, every rank iterates on its rows: the one-process partition
        np.random.seed(11)
        labels = clustering_rph_kmeans_sharded(mine, x if rank == 0 else None, 6, n_init=3, device="cpu")
        if rank == 0:
            np.random.seed(11)
            want = RPHKMeans(n_init=3, n_clusters=6, device="cpu").fit_predict(x)
            assert labels.dtype == np.int32 and np.array_equal(labels, want)
        else:
            assert labels is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_lloyd_gives_the_one_process_partition(world):
    _spawn(_lloyd_worker, world)


def _control_worker(rank, world, port):
    """the data group's timeout is 2 s; rank 0 is busy for 5 s (training, cache files, assembly in the real run): the other ranks
    wait for it on the control plane and nothing times out"""
    import datetime
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=2))
    try:
        assert pdist.control_group() is not None
        t0 = time.time()
        if rank == 0:
            time.sleep(5)
        got = pdist.agreed({"weights": 42} if rank == 0 else None)
        assert got == {"weights": 42} and (rank == 0 or time.time() - t0 > 4)
        if rank == 0:
            time.sleep(3)
        assert pdist.everyone(True)                 # (routed over the control plane once it exists)
        if rank == 0:
            time.sleep(3)
        pdist.wait_for_all()
        t = torch.ones(1)
        dist.all_reduce(t)                          # the data group still works: nothing was left pending on it
        assert int(t.item()) == world
    finally:
        pdist.leave()
    assert not dist.is_initialized() and pdist._CONTROL is None


def test_waits_for_rank0_use_the_control_plane():
    _spawn(_control_worker, 2)


# ------------------------------------------------------------------ the super-k-mer form on N > 1 ranks (dist.MiniSharded)


def _mini_cfg():
    return synth.SynthConfig(n_pairs=24_000, n_barcodes=150, n_genomes=3, genome_len=40_000, fragment=10_000, sub_rate=0.01, n_rate=0.05, seed=321)


def _shard_by_runs(s, rank, world):
    """this rank's run range of a device stream, as a stream of its own (word aligned; what ``dist.shard_stream`` does on the host)"""
    return pdist.shard_stream(ReadStream(s.codes.cpu(), s.valid.cpu(), s.n_chars, s.run_off, s.run_names), rank, world).to("cuda:0")


def _mini_sharded_worker(rank, world, port, outdir, backend, saturate):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if saturate:
            # 3.1 M copies of one 21-mer on EVERY rank: every part arrives saturated, the sum must stay at 2^21 and must not carry
            # into the code; a tandem repeat and random text besides
            rng = np.random.RandomState(3)
            rnd = bytes(rng.choice(list(b"ACGT"), size=60_000).astype(np.uint8))
            s = ReadStream.from_runs([("a", b"A" * 1_600_000 + b"N" + b"T" * 1_500_040 + b"N"), ("b", b"ACG" * 30_000 + b"N"),
                                      ("c", rnd + b"N")], device="cuda:0")
            part = s
        else:
            s = synth.generate(_mini_cfg(), device="cuda:0")
            part = _shard_by_runs(s, rank, world)
        rows = part.rows(2000 if not saturate else 0)
        plan = kmer.Plan(rows, "cuda:0")
        tnf, abd, ms = pdist.features_sharded_mini(part, plan, 21, 4, 10, 400)
        assert ms.local.n_buckets == ms.union.n_buckets >= 512 and ms.local.log2_bucket <= ms.union.log2_bucket
        c, n = ms.owned_items()
        np.savez(os.path.join(outdir, f"m{rank}.npz"), c=c, n=n, tnf=tnf.cpu().numpy(), abd=abd.cpu().numpy(),
                 names=np.array(rows.names), sent=ms.bytes_sent)
        # counting again with the same object (what every bench step does) gives the same rows
        ms.count(part, plan)
        _, abd2 = kmer.features(part, plan, k_tnf=None, table=ms.local, window=10, vsize=400)
        assert torch.equal(abd2, abd)
        # the exchange keeps the part size of the first batch and does not ask the device again: a batch that does not fit it is
        # not exchanged, the status word says so on every rank, and count() sizes and counts again
        kept = ms._cap1
        ms._cap1 = 8
        ms.count(part, plan)
        assert ms._cap1 == kept
        _, abd3 = kmer.features(part, plan, k_tnf=None, table=ms.local, window=10, vsize=400)
        assert torch.equal(abd3, abd)
    finally:
        dist.destroy_process_group()


def _check_mini_sharded(tmp_path, world, backend="gloo", saturate=False):
    _spawn(_mini_sharded_worker, world, str(tmp_path), backend, saturate)
    parts = [np.load(str(tmp_path / f"m{r}.npz")) for r in range(world)]
    if saturate:
        rng = np.random.RandomState(3)
        rnd = bytes(rng.choice(list(b"ACGT"), size=60_000).astype(np.uint8))
        s = ReadStream.from_runs([("a", b"A" * 1_600_000 + b"N" + b"T" * 1_500_040 + b"N"), ("b", b"ACG" * 30_000 + b"N"), ("c", rnd + b"N")], device="cuda:0")
        text = s.decode()
        otab = oracle.Table(21, threads=4)
        for _ in range(world):
            otab.count(text)                                 # every rank holds a copy of the same reads
    else:
        s = synth.generate(_mini_cfg(), device="cuda:0")
        text = s.decode()
        otab = oracle.Table(21, threads=4).count(text)
    # the owners' ranges together are the oracle's table (counts saturate at 2^21 exactly as one rank's table would)
    codes = np.concatenate([p["c"] for p in parts]); counts = np.concatenate([p["n"] for p in parts])
    order = np.argsort(codes)
    ocodes, ocounts = otab.items()
    assert np.array_equal(codes[order], ocodes) and np.array_equal(counts[order], np.minimum(ocounts, 1 << 21))
    if saturate:
        assert counts.max() == 1 << 21
        rows = s.rows(0)
        for p in parts:                                       # every rank has the rows of the whole text, looked up in the summed table
            for r in range(len(rows)):
                assert np.array_equal(p["abd"][r], oracle.abd_row(text[rows.start[r]:rows.end[r]], 21, otab, 10, 400))
        return
    # the ranks' rows, in rank order, are the rows of the whole file
    rows = s.rows(2000)
    names = [n for p in parts for n in p["names"].tolist()]
    assert names == list(rows.names)
    abd = np.concatenate([p["abd"] for p in parts]); tnf = np.concatenate([p["tnf"] for p in parts])
    plan = kmer.Plan(rows, "cuda:0")
    one = kmer.count_kmers(s, 21, rows=plan, emit=(10, 400))
    want_tnf, want_abd = kmer.features(s, plan, k_tnf=4, table=one, window=10, vsize=400)
    assert np.array_equal(abd, want_abd.cpu().numpy()) and np.array_equal(tnf, want_tnf.cpu().numpy())
    for r in range(0, len(rows), max(1, len(rows) // 8)):
        assert np.array_equal(abd[r], oracle.abd_row(text[rows.start[r]:rows.end[r]], 21, otab, 10, 400))
    assert world == 1 or all(int(p["sent"]) > 0 for p in parts)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 4])
def test_super_kmer_form_on_several_ranks(tmp_path, world):
    """every rank counts ITS runs with the one-GPU pipeline up to the provisional words, entries go to bucket-range owners (8 bytes per
    distinct k-mer), merged bins come back (2 bytes), the lookup half finishes: rows == the one-process rows == the oracle's, the
    owners' table ranges together == the oracle's table (3 ranks: uneven ranges)"""
    _check_mini_sharded(tmp_path, world)


@pytest.mark.gpu
def test_super_kmer_form_saturating_counts_on_two_ranks(tmp_path):
    _check_mini_sharded(tmp_path, 2, saturate=True)


@pytest.mark.gpu
def test_super_kmer_form_over_a_one_rank_rccl_group(tmp_path):
    """the same exchange with RCCL's collectives (a one-GPU box holds a one-rank group): all-gather of the fills, both all-to-alls"""
    _check_mini_sharded(tmp_path, 1, backend="nccl")


@pytest.mark.gpu
def test_super_kmer_form_through_the_checked_build():
    """the N-rank kernels (count half's entries and occupancy, entry gather, owner merge and bins, lookup half) with every global
    store checked against its buffer (PANGAEA_LIB=checked: a wrong index sets PG_STATUS_BOUNDS, which MiniSharded raises on every
    rank) -- two ranks sharing the GPU and the one-rank RCCL group"""
    import subprocess
    import sys
    from .conftest import ROOT
    env = dict(os.environ, PANGAEA_LIB="checked")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_dist_gloo.py"),
                        "-k", "(super_kmer_form_on_several_ranks and 2) or one_rank_rccl or saturating_counts"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "no tests ran" not in r.stdout
