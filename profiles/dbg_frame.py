"""One frame against the oracle, default and gf_exact=1:  python profiles/dbg_frame.py frame.npy [strategy]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import underwater_image_enhancement_amd as uw
from oracle import uwie_oracle as orc

u8 = np.load(sys.argv[1])
st = int(sys.argv[2]) if len(sys.argv) > 2 else 2
want = orc.enhance_u8(u8, st)
for name, kw in (("default", {}), ("gf_exact", {"gf_exact": 1})):
    got = uw.enhance(u8, strategy=st, **kw)
    d = got.astype(int) - want.astype(int)
    idx = np.argwhere(d != 0)
    print(name, "differing bytes", len(idx), "max", np.abs(d).max())
    for y, x, c in idx[:5]:
        print("   at", y, x, c, "got", got[y, x, c], "want", want[y, x, c], "input px", u8[y, x])
