"""-m gpu: the in-launch finalize roles of the fused surface scan under conditions the host's residency argument does not
cover by itself (round-4 review): a process confined to HALF and to an EIGHTH of the chip's CUs (ROC_GLOBAL_CU_MASK: the
roles and the tiles they wait for then share far fewer slots), and a second process saturating the device meanwhile.
Each runs the config-5 shard's surface step in a child process and compares loss and gradient, bit for bit, with the same
step taken with the roles off (the stand-alone finalize launch): equal, finite, and no role gave up."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from geometrics_amd import meshgen, ops, utils
dev = torch.device("cuda:0")
V, Fc = meshgen.icosphere(4)
B, num = 8, 3000
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
verts = to(meshgen.jittered_batch(V, B)).requires_grad_(True)
faces, gt = to(Fc), to(meshgen.gt_cloud(B, num))
gi = ops.GtIndex(gt)
out = []
for tail in (True, False):
    ops.scan_finalize_tail = tail
    for rep in range(int(sys.argv[1])):
        verts.grad = None
        ops.manual_seed(5 + rep)
        loss = utils.batch_point_to_surface(verts, {"faces": faces}, gt, num=num, gt_index=gi)
        loss.backward()
        out.append((loss.detach().clone(), verts.grad.clone()))
torch.cuda.synchronize()
n = len(out) // 2
ok = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and bool(torch.isfinite(a[0])) and bool(torch.isfinite(a[1]).all())
         for a, b in zip(out[:n], out[n:]))
print("CUS", torch.cuda.get_device_properties(0).multi_processor_count, "OK" if ok and not ops.finalize_roles_gave_up() else "MISMATCH")
""" % ROOT

HOG = r"""
import time, torch
a = torch.randn(8192, 8192, device="cuda")
t_end = time.time() + float(%s)
while time.time() < t_end:
    for _ in range(4):
        torch.mm(a, a)
    torch.cuda.synchronize()
"""


def _run(env_extra, reps=3):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD, str(reps)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("CUS")][-1]
    assert line.endswith("OK"), line
    return int(line.split()[1])


@pytest.mark.parametrize("bits", [128, 32])
def test_roles_with_a_fraction_of_the_cus(gpu, bits):
    """ROC_GLOBAL_CU_MASK with `bits` of the 256 CUs enabled: what a CU-partitioned tenant sees."""
    cus = _run({"ROC_GLOBAL_CU_MASK": hex((1 << bits) - 1)})
    assert cus > 0


def test_roles_beside_a_process_that_saturates_the_device(gpu):
    hog = subprocess.Popen([sys.executable, "-c", HOG % "40"])
    try:
        import time
        time.sleep(8)              # its import + first products
        _run({}, reps=6)
    finally:
        hog.kill()
        hog.wait()
