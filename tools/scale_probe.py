"""Developer probe: stage timings vs number of blocks in one batch (same data, growing prefix)."""
import bz2, os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen
import indexed_bzip2_amd as m

def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "text"
    counts = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [32, 64, 128, 256, 512, 1024, 2048]
    piece = 9_000_000
    base = datagen.text_like(piece, 7) if kind == "text" else datagen.random_bytes(piece, 7)
    n = (max(counts) + 9) // 10 + 1
    def comp(i):
        return bz2.compress(bytes([i & 255, (i >> 8) & 255]) * 8 + base[16:], 9)
    with ThreadPoolExecutor(32) as ex:
        enc = b"".join(ex.map(comp, range(n)))
    offs = m.find_magic(enc)
    dec = m.Decoder(); dec.set_input(enc)
    for c in counts:
        sub = offs[:c]
        dec.decode_batch(sub)
        res, total = dec.decode_batch(sub)
        t = dec.timings()
        ks = " ".join(f"{k}={v:.2f}" for k, v in t["kernels"].items())
        print(f"blocks={len(sub)} MB={total/1e6:.0f} total_ms={t['ms_kernel_sum']:.2f} {ks}", flush=True)

if __name__ == "__main__":
    main()
