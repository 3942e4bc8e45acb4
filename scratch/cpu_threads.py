import torch, time, torch.nn.functional as F, os
x = torch.randn(1, 256, 200, 336); w = torch.randn(256, 256, 3, 3)
for nt in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(nt)
    F.conv2d(x, w, padding=1)
    t0 = time.time(); F.conv2d(x, w, padding=1); F.conv2d(x, w, padding=1); print(nt, (time.time() - t0) / 2)
