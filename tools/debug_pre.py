import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centernet_amd import native, image as I
from oracle import pre_oracle as P
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_pre import _warp_norm, _img, MEAN, STD
dev = torch.device("cuda:0")
(h, w), (oh, ow), m = (20, 28), (32, 40), [0.7, 0.05, -1.3, -0.04, 0.66, 2.2]
img = _img(h, w, 3)
got = _warp_norm(dev, img, m, oh, ow, False)
u8 = P.warp_bilinear_u8(img, m, (ow, oh))
ref = I.normalize_chw(u8, MEAN, STD)[None]
d = np.argwhere(got != ref)
print("ndiff", len(d), "of", got.size)
mean = np.float32(MEAN).astype(np.float64); std = np.float32(STD).astype(np.float64)
for b, c, y, x in d[:10]:
    lv = (got[b, c, y, x].astype(np.float64) * std[c] + mean[c]) * 255
    print((c, y, x), got[b, c, y, x], ref[b, c, y, x], "level got %.4f ref %d" % (lv, u8[y, x, c]))
    sx = (m[0] * x + m[1] * y) + m[2]; sy = (m[3] * x + m[4] * y) + m[5]
    print("   sx,sy", repr(sx), repr(sy))
