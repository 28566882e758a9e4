"""The N > 1 path on CPU: two gloo ranks, contiguous rid shards, host-side concatenation, and the
barrier / max-over-ranks plumbing bench.py uses.  There is no GPU here, so each rank's shard result is
produced by the checker (oracle) standing in for the device; what is under test is the product's sharding
(`shard_ranges`), merge (`concat_shards`) and `sctagger_amd.shard` reductions."""
import os
import socket
import subprocess
import sys
import textwrap

from tests import util

ROOT = os.path.dirname(util.HERE)

WORKER = textwrap.dedent("""
    import sys, numpy as np
    sys.path.insert(0, %r)
    from sctagger_amd import shard, synth
    from sctagger_amd.match_trie import shard_ranges, concat_shards
    from oracle import trie_oracle
    rank, local_rank, world = shard.rank_info()
    dist = shard.init_dist()
    assert dist is not None and dist.get_world_size() == 2 and dist.get_backend() == "gloo"
    bc = synth.make_barcodes(200, seed=5)
    segs, offs = synth.make_segments(3001, bc, seed=5)          # odd count: uneven shards
    lo, hi = shard_ranges(offs.size - 1, world)[rank]
    mine = trie_oracle.match(bc, segs[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo], 2, threads=16)
    shard.barrier(dist)
    t, units = shard.reduce_timing(dist, 1.0 + rank, hi - lo)
    assert t == 2.0 and units == 3001.0, (t, units)
    assert shard.gather_objects(dist, {"rank": rank}) == [{"rank": 0}, {"rank": 1}]
    parts = [None, None] if rank == 0 else None
    dist.gather_object(mine, parts, dst=0)
    if rank == 0:
        whole = trie_oracle.match(bc, segs, offs, 2, threads=16)
        got = concat_shards(parts)
        for a, b in zip(got, whole):
            assert np.array_equal(a, b)
        print("DIST-OK")
    shard.barrier(dist)
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_gloo_shards(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-2000:]
    assert b"DIST-OK" in outs[0][0]


import json

import pytest


@pytest.mark.gpu
def test_two_ranks_on_the_device(tmp_path):
    """bench.py's N = 2 path with the DEVICE in both ranks (they share GPU 0 on a one-GPU box): torchrun launch, gloo
    barrier / reductions, one context per process, every rank its own seeded shard; the line carries both ranks' times."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--segments", "300000"], capture_output=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    line = [l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["value_resident"] > 0
    assert [r["rank"] for r in d["per_rank"]] == [0, 1] and all(r["e2e_ms_per_step"] > 0 for r in d["per_rank"])
    assert d["config"]["paths"]["filter"] == 300000
