"""Developer probe: PCIe-inclusive rate of the batch API -- compressed bytes start in host memory, decoded bytes end in
(page-locked) host memory: set_input (H2D) + decode_batch + copy_output (D2H)."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench
import indexed_bzip2_amd as m


def main():
    path, enc, meta = bench.build_workload(2 * 1024**3, 214_748_364, "/tmp/indexed_bzip2_amd_bench", 0, 1, lambda: None)
    offsets = meta["offsets"]
    n = len(offsets)
    dec = m.Decoder(device=0, max_batch_blocks=n)
    host_in = torch.frombuffer(bytearray(enc), dtype=torch.uint8).pin_memory()
    host_out = torch.empty(meta["decoded_bytes"], dtype=torch.uint8).pin_memory()
    offs_c, res_c = dec.make_arrays(offsets)
    L = m.lib()
    for it in range(4):
        t0 = time.perf_counter()
        rc = L.mi355x_bz2_set_input_host(dec._h, ctypes.cast(host_in.data_ptr(), ctypes.c_char_p), len(enc))
        assert rc == 0
        t1 = time.perf_counter()
        total = dec.decode_batch_into(offs_c, n, res_c)
        t2 = time.perf_counter()
        rc = L.mi355x_bz2_copy_output(dec._h, 0, total, ctypes.c_void_p(host_out.data_ptr()))
        assert rc == 0
        t3 = time.perf_counter()
        print(f"pass {it}: H2D+swap {1e3 * (t1 - t0):.1f} ms, decode {1e3 * (t2 - t1):.1f} ms, D2H {1e3 * (t3 - t2):.1f} ms, "
              f"total {1e3 * (t3 - t0):.1f} ms = {total / (t3 - t0) / 1e6:.0f} MB/s decoded", flush=True)
    assert bytes(host_out[:16].numpy()) == dec.copy_output(0, 16)


if __name__ == "__main__":
    main()
