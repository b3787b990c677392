import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import underwater_image_enhancement_amd as uw
from oracle import uwie_oracle as orc
from test_gpu_fuzz import random_frame
rng = np.random.default_rng(3)
names = ["strong_dehazing", "medium_dehazing", "light_enhancement", "clahe_enhancement", "histogram_equalization"]
worst = 0.0; nd = 0; n = 0; byte_diff = 0
for i in range(120):
    u8 = random_frame(rng); name = names[rng.integers(5)]
    x = orc.normalise_u8(u8)
    want = orc.DictStrategyOracle.run(x, name, {})
    got = np.asarray(uw.EnhancementStrategies.apply_strategy(x, name, {}))
    d = np.abs(got - want).max()
    worst = max(worst, d); n += 1; nd += d > 0
    byte_diff += int(np.count_nonzero((got * 255).astype(np.uint8) != (want * 255).astype(np.uint8)))
print("cases", n, "differing", nd, "max abs diff", worst, "bytes differing after (x*255).astype(u8):", byte_diff)
