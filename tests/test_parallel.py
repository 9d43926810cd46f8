"""File-shard helpers and the N>1 path on CPU: two gloo ranks each own a shard, compute it,
and every rank assembles the full result without any data-path collective."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from audio_feature_extraction_amd.parallel import lpt_partition, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lpt_partition_balances_and_covers():
    lengths = [100, 5, 90, 20, 60, 60, 1, 300]
    parts = lpt_partition(lengths, 3)
    assert sorted(i for p in parts for i in p) == list(range(len(lengths)))
    loads = [sum(lengths[i] for i in p) for p in parts]
    assert max(loads) == 300 and max(loads) - min(loads) <= 150
    assert all(p == sorted(p) for p in parts)
    eq = lpt_partition([10] * 8, 4)
    assert sorted(len(p) for p in eq) == [2, 2, 2, 2]
    assert lpt_partition([], 2) == [[], []]
    assert lpt_partition([3, 4], 1) == [[0, 1]]


def test_shard_range_is_a_partition():
    for n in (0, 1, 7, 8, 1000, 8001):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, {root!r})
    from audio_feature_extraction_amd.parallel import shard_range, gather_shards
    from audio_feature_extraction_amd.synth import make_clip
    from oracle import cpu_ref as R            # stand-in compute: this test runs without a GPU

    def rows(lo, hi):
        out = []
        for i in range(lo, hi):
            s = R.extract_stats(make_clip(i, 8000, 0.5), sr=8000, frame_length=256, hop_length=64, n_mfcc=5)
            out.append(np.concatenate([s["mfcc_mean"], s["mfcc_std"], [s["energy_mean"]]]).astype(np.float32))
        return np.stack(out) if out else np.zeros((0, 11), np.float32)

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 7
    lo, hi = shard_range(n, rank, world)
    full = gather_shards(rows(lo, hi), n, rank, world)
    dist.barrier()
    if rank == 0:
        np.save({out!r}, full)
    dist.destroy_process_group()
""")


def test_two_rank_gloo_file_shard(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "gathered.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(2)]
    assert [p.wait(timeout=240) for p in procs] == [0, 0]
    got = np.load(out)
    from audio_feature_extraction_amd.synth import make_clip
    from oracle import cpu_ref as R
    ref = []
    for i in range(7):
        st = R.extract_stats(make_clip(i, 8000, 0.5), sr=8000, frame_length=256, hop_length=64, n_mfcc=5)
        ref.append(np.concatenate([st["mfcc_mean"], st["mfcc_std"], [st["energy_mean"]]]).astype(np.float32))
    np.testing.assert_array_equal(got, np.stack(ref))      # same rows, global order, bit for bit
