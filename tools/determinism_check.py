import numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sctagger_amd import _native, synth
import bench
bc = synth.make_barcodes(25000, seed=3)
segs, offs = bench.make_batch(10_000_000, bc, 3)
m = _native.Matcher(0); m.set_barcodes(bc); m.upload_segments(segs, offs)
ref = None
for i in range(4):
    m.run(2); r = m.download()
    if ref is None: ref = [x.copy() for x in r]
    else:
        assert all(np.array_equal(a, b) for a, b in zip(ref, r)), "run %d differs" % i
print("4 runs identical:", ref[2].size, "hits")
