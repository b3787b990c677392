import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import underwater_image_enhancement_amd as uw
from oracle import uwie_oracle as orc
rng = np.random.default_rng(20240516)
x64 = rng.random((256, 256, 3))
ES = orc.DictStrategyOracle
for name in ("strong_dehazing", "medium_dehazing", "light_enhancement"):
    for params in ({"apply_gamma": False}, {"apply_gamma": True}, orc.CONFIG_STRATEGIES[name]):
        want = ES.run(x64, name, params)
        got = uw.EnhancementStrategies.apply_strategy(x64, name, params)
        d = np.abs(got - want)
        i = np.unravel_index(np.argmax(d), d.shape)
        print(name, params, "max err %.3e at %s got %.17g want %.17g; n>1e-12: %d" % (d.max(), i, got[i], want[i], (d > 1e-12).sum()))
