import sys, os
sys.path.insert(0, "/root/repo")
import torch
dev = torch.device("cuda:0")
tun = torch.cuda.tunable
mode = sys.argv[1]
print("mode", mode, flush=True)
tun.enable(True)
tun.tuning_enable(True)
if mode == "full":
    tun.record_untuned_enable(False)
    tun.set_max_tuning_duration(30)
    tun.set_max_tuning_iterations(30)
if mode in ("full", "file"):
    import tempfile
    tun.set_filename(os.path.join(tempfile.mkdtemp(), "s.csv"))
    if hasattr(tun, "write_file_on_exit"):
        tun.write_file_on_exit(False)
x = torch.randn(8, 2562, 963, device=dev)
w = torch.randn(963, 192, device=dev)
y = torch.matmul(x, w)
g = torch.randn(8 * 2562, 192, device=dev)
dx = torch.matmul(g, w.t())
torch.cuda.synchronize()
tun.tuning_enable(False)
print("ok", tun.get_results()[:2] if hasattr(tun, "get_results") else "", flush=True)
