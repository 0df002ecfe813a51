"""Developer probe: k_hscan on blocks with few groups = the cost of parsing a block's header, selectors and trees."""
import bz2, os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from indexed_bzip2_amd import _native as nat
import datagen
for name, raw in (("text 20 kB", datagen.text_like(20000, 3)), ("random 20 kB", random.Random(2).randbytes(20000)),
                  ("text 200 kB", datagen.text_like(200000, 3)), ("random 200 kB", random.Random(2).randbytes(200000))):
    data = bz2.compress(raw, 9)
    d = nat.Decoder(); d.set_input(data)
    for _ in range(3):
        res, total = d.decode_batch([32])
    t = d.timings()["kernels"]
    print(f"{name}: k_hscan {t['k_hscan']:.3f} ms, k_hsym {t['k_hsym']:.3f}, k_mtf {t['k_mtf<144>'] + t['k_mtf<272>']:.3f}, pipeline {d.pipeline_ms():.3f} ms", flush=True)
    d.close()
