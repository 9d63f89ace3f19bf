"""How much of a host-to-device and a device-to-host copy overlap on this box (page-locked memory, two streams).
usage: python scripts/pcie_duplex.py [MB per direction] [pieces]"""
import sys, time
import torch
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pieces = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n = mb * 1024 * 1024 // 8
h_up = torch.empty(n, dtype=torch.float64).pin_memory(); h_dn = torch.empty(n, dtype=torch.float64).pin_memory()
d_up = torch.empty(n, dtype=torch.float64, device="cuda"); d_dn = torch.zeros(n, dtype=torch.float64, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(up, dn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c = n // pieces
    for k in range(pieces):
        if up:
            with torch.cuda.stream(s1): d_up[k * c:(k + 1) * c].copy_(h_up[k * c:(k + 1) * c], non_blocking=True)
        if dn:
            with torch.cuda.stream(s2): h_dn[k * c:(k + 1) * c].copy_(d_dn[k * c:(k + 1) * c], non_blocking=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for _ in range(2): run(True, True)
for name, a, b in (("up", True, False), ("down", False, True), ("both", True, True)):
    ts = sorted(run(a, b) for _ in range(7))
    print("%s: %d MB per direction in %d pieces: %.2f ms (median of 7) = %.1f GB/s per direction" % (name, mb, pieces, ts[3], mb / 1024 / ts[3] * 1e3))
