import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cc_amd._lib import engine, STREAM
E = engine()
B, M, Cin, H, W = 4, 16, 16, 256, 832
a = torch.randn(B, M, H, W, device="cuda"); x = torch.randn(B, Cin, H, W, device="cuda")
gw = torch.zeros(M, Cin, 3, 3, device="cuda")
ws = torch.empty(E.call("cc_conv2d_wgrad_ws_bytes", B, M, H, W, Cin, 3, 3, 1) // 4 + 64, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    E.call("cc_conv2d_wgrad", a, x, gw, ws, B, M, H, W, M * H * W, Cin, H, W, Cin * H * W, 3, 3, 1, 1, Cin * 9, 9, 0, STREAM)
torch.cuda.synchronize()
