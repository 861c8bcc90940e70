"""Random soak of the decode / 2:1 / any-ratio paths against the oracle (runs on the GPU box):
    python tools/soak.py [seed] [cases]
Fresh seeds every time it is used; the committed tests hold the fixed-seed fuzz."""
import os
import sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import gpu_helpers as gh
from oracle_lib import Oracle
import metalbt709decoder_amd as mb
from metalbt709decoder_amd import _capi
oracle = Oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ctx = gh.context()
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    kind = int(rng.integers(0, 3))
    gamma = int(rng.integers(0, 4))
    w = 4 * int(rng.integers(1, 600)); h = 4 * int(rng.integers(1, 40))
    y, c = gh.random_nv12(w, h, seed=int(rng.integers(0, 1 << 30)))
    if kind == 0:
        a = rng.integers(0, 256, (h, w), dtype=np.uint8) if rng.integers(0, 3) == 0 else None
        got = gh.gpu_decode(y, c, gamma if a is None else mb.MetalBT709GammaSRGB, alpha=a)
        want = oracle.decode_nv12(gamma if a is None else mb.MetalBT709GammaSRGB, y, c, alpha=a)
    elif kind == 1:
        dec = gh.make_decoder(gamma, options={_capi.OPT_HALF_KERNEL: int(rng.integers(0, 2)),
                                              _capi.OPT_HALF_WORKGROUPS: int(rng.integers(1, 400))})
        got = gh.gpu_decode_half(y, c, gamma, decoder=dec); want = oracle.decode_nv12_half(gamma, y, c)
    else:
        ow, oh = int(rng.integers(1, 2 * w)), int(rng.integers(1, 2 * h))
        dec = gh.make_decoder(gamma); buf = gh.make_buffer(y, c, dec.gamma); tex = ctx.makeBGRATexture((ow, oh))
        assert dec.decodeBT709Scaled(buf, tex, None, True)
        got = ctx.getBGRATexturePixels(tex).view(np.uint8).reshape(oh, ow * 4); want = oracle.decode_nv12_scaled(gamma, y, c, ow, oh)
    if not np.array_equal(got, want):
        bad += 1; print("MISMATCH", case, kind, gamma, w, h)
print("soak done, mismatches:", bad)
