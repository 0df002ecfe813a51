"""Developer probe: kernel times per decoded GB at different bzip2 levels (block sizes), same data."""
import bz2, os, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen
import indexed_bzip2_amd as m

def main():
    os.environ["MI355X_BZ2_NO_SPLIT"] = "1"
    base = datagen.text_like(9_000_000, 7)
    for level in (9, 5, 4, 2):
        def comp(i):
            return bz2.compress(bytes([i & 255, (i >> 8) & 255]) * 8 + base[16:], level)
        with ThreadPoolExecutor(32) as ex:
            enc = b"".join(ex.map(comp, range(110)))
        offs = m.find_magic(enc)
        dec = m.Decoder(); dec.set_input(enc)
        dec.decode_batch(offs)
        res, total = dec.decode_batch(offs)
        t = dec.timings()
        gb = total / 1e9
        print(f"level {level}: {len(offs)} blocks, {total/1e6:.0f} MB: " + " ".join(f"{k}={v/gb:.2f}" for k, v in t["kernels"].items()) + "  (ms per decoded GB)", flush=True)
        dec.close()

if __name__ == "__main__":
    main()
