"""Multi-GPU plumbing for the match_trie path: one process per GPU, independent rid shards, no data-path
collective.  torch.distributed is imported only when WORLD_SIZE > 1 and only for the
barrier and two scalar reductions of the benchmark (gloo: CPU scalars; nothing moves over xGMI)."""
import os
import socket
import subprocess
import sys


def launch_ranks(n, script, argv, env=None, timeout=None):
    """Start `n` ranks of `script argv...` as CHILD processes of this one (the reference's `Pool(threads)` fan-out,
    /root/reference/scTagger.py:703-704, as one process per GPU) and wait for them: RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT in each child's environment as torchrun would set them, stdout and stderr inherited (rank 0
    prints the result line).  The caller must not have touched the GPU: it only waits.  Nothing is exec'ed in place.
    -> the first non-zero exit status of a rank (the others are then terminated), or 0."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=dict(base, RANK=str(r), LOCAL_RANK=str(r))))
    import time
    deadline = None if timeout is None else time.monotonic() + timeout
    status = 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and status == 0:
                status = rc
        if status != 0 or (deadline is not None and time.monotonic() > deadline):
            for p in live:            # a rank failed (or time ran out): the others would wait in the barrier for ever
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            return status or 124
        if live:
            time.sleep(0.02)
    return status


def rank_info():
    """-> (rank, local_rank, world_size) from the torchrun environment (1 process if unset)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_dist():
    """Initialise the process group if launched with WORLD_SIZE > 1; returns torch.distributed or None."""
    _, _, world = rank_info()
    if world <= 1:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    return dist


def barrier(dist):
    if dist is not None:
        dist.barrier()


def reduce_timing(dist, elapsed_s, units):
    """-> (max elapsed over ranks, total units over ranks)."""
    if dist is None:
        return float(elapsed_s), float(units)
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    u = torch.tensor([float(units)], dtype=torch.float64)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t[0]), float(u[0])


def gather_objects(dist, obj):
    """-> list with every rank's `obj` (on every rank); [obj] for a single process."""
    if dist is None:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def bind_to_device_numa(device):
    """Best effort: restrict this process to the CPUs of the NUMA node the GPU hangs off, BEFORE it allocates page-locked
    buffers (they are placed on the node of the allocating thread).  With eight ranks streaming 57 GB/s each out of host
    memory, buffers on the far socket would cross the inter-socket link.  -> the node number or None."""
    from . import _native
    try:
        bus = _native.device_pci_bus_id(device)
        if not bus:
            return None
        with open("/sys/bus/pci/devices/%s/numa_node" % bus) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            cpus = set()
            for part in f.read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        allowed = os.sched_getaffinity(0) & cpus
        if allowed:
            os.sched_setaffinity(0, allowed)
            return node
    except Exception:
        pass
    return None
