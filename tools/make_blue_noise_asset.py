"""Packs the blue-noise sampler tables the reference embeds as C arrays
(src/utils/blue_noise_sampler.hpp: sobol_256spp_256d[256*256], scramblingTile[128*128*8],
rankingTile[128*128*8]; Heitz et al. 2019, "A Low-Discrepancy Sampler that Distributes Monte
Carlo Errors as a Blue Noise in Screen Space") into assets/blue_noise/heitz2019_256spp_256d.bin:
three uint8 tables back to back (every value is in 0..255).  Data asset, like the env map."""
import os, re, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open("/root/reference/src/utils/blue_noise_sampler.hpp").read()
out = []
for name, n in (("sobol_256spp_256d", 65536), ("scramblingTile", 131072), ("rankingTile", 131072)):
    m = re.search(r"static const int %s\[.*?\]\s*=\s*\{(.*?)\};" % name, txt, flags=re.S)
    a = np.array([int(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()], dtype=np.int64)
    assert len(a) == n and a.min() >= 0 and a.max() <= 255, name
    out.append(a.astype(np.uint8))
path = os.path.join(ROOT, "assets", "blue_noise", "heitz2019_256spp_256d.bin")
np.concatenate(out).tofile(path)
print(path, os.path.getsize(path))
