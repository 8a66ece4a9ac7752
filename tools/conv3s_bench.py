import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from custom_d_fine_amd import hip as H
dev = torch.device("cuda", 0)
for cin, cout, side in [(32, 32, 160), (16, 32, 160), (32, 32, 80)]:
    x = torch.randn(32, cin, side, side, device=dev).bfloat16()
    w2 = H.conv_pack_weights(torch.randn(cout, cin, 3, 3, device=dev), False)
    for _ in range(3): H.conv_forward_bf16(x, w2, cout, 3)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): H.conv_forward_bf16(x, w2, cout, 3)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    io = 2.0 * 32 * side * side * (cin + cout)
    print(f"{cin}->{cout} @{side}: {us:7.1f} us  {io/us/1e3:7.1f} GB/s  (DFINE_CONV3X3_ROWS32={os.environ.get('DFINE_CONV3X3_ROWS32','1')})")
