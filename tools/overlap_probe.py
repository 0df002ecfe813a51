"""Developer probe: two decoder contexts driven from two host threads (double buffering across steps)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import indexed_bzip2_amd as m


def main():
    nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    path, enc, meta = bench.build_workload(2 * 1024**3, 214_748_364, "/tmp/indexed_bzip2_amd_bench", 0, 1, lambda: None)
    offsets = meta["offsets"]
    d_in = torch.frombuffer(bytearray(enc), dtype=torch.uint8).cuda()
    decs = []
    for _ in range(nctx):
        d = m.Decoder(device=0, max_batch_blocks=len(offsets))
        d.set_input_device(d_in.data_ptr(), len(enc), keepalive=d_in)
        res, total = d.decode_batch(offsets)
        assert total == meta["decoded_bytes"] and all(r["status"] == 0 for r in res)
        decs.append(d)
    torch.cuda.synchronize()

    def worker(d, k):
        for _ in range(k):
            res, total = d.decode_batch(offsets)
            assert total == meta["decoded_bytes"]

    t0 = time.perf_counter()
    threads = [threading.Thread(target=worker, args=(d, steps // nctx)) for d in decs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    done = (steps // nctx) * nctx
    print(f"contexts={nctx} steps={done}: {dt / done * 1e3:.1f} ms/step, {meta['decoded_bytes'] * done / dt / 1e6:.0f} MB/s "
          f"(GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')})", flush=True)


if __name__ == "__main__":
    main()
