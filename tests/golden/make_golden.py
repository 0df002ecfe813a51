"""Generate tests/golden/reference_vectors.json from the REAL reference (oracle/_ref/ref_bz2, compiled from
/root/reference/src by oracle/Makefile).  Run in the authoring container only; the JSON (data: inputs are reproducible
from tests/datagen.py + CPython bz2, outputs are the reference's) is committed, the reference itself never travels.

For every case: sha256 of the compressed input, the block map of ParallelBZ2Reader::blockOffsets(), the serial
BZ2Reader map, and per data block (offset, encodedSize, headerCRC, calculatedCRC, decodedSize, fnv64(data)).
Also records the reference's verdict (exception type) for a set of corrupted / truncated inputs.
"""
import hashlib
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen
from oracle import oracle as O

OUT = os.path.join(ROOT, "tests", "golden", "reference_vectors.json")
FIX = os.path.join(ROOT, "tests", "golden", "fixtures")


def ref_case(path):
    m = [[int(a), int(b)] for a, b in (l.split() for l in O.ref_run("map", path, 2).strip().splitlines())]
    sm = [[int(a), int(b)] for a, b in (l.split() for l in O.ref_run("smap", path).strip().splitlines())]
    blocks = []
    for l in O.ref_run("blocks", path).strip().splitlines():
        f = l.split()
        blocks.append([int(f[0]), int(f[1]), int(f[2], 16), int(f[3], 16), int(f[4]), f[5]])
    return {"map": m, "serial_map": sm, "blocks": blocks}


def corrupt_cases():
    raw = datagen.text_like(200_000, 41)
    enc = datagen.compress(raw, 9)
    cases = {}
    for flip in (len(enc) // 2, len(enc) // 3, 40, 60, 100, 2000, 14, 15, 16, 20, 30):
        b = bytearray(enc)
        b[flip] ^= 0x10
        cases[f"flip-{flip}"] = bytes(b)
    for cut in (len(enc) - 11, len(enc) // 2, 100, 20, 9):
        cases[f"cut-{cut}"] = enc[:cut]
    return cases


def main():
    assert O.ref_available(), "build oracle/_ref first: make -C oracle ref"
    golden = {"fixtures": {}, "generated": {}, "multistream": {}, "probes": {}, "damaged": {}}
    for f in sorted(os.listdir(FIX)):
        if f.endswith(".bz2"):
            p = os.path.join(FIX, f)
            data = open(p, "rb").read()
            golden["fixtures"][f[:-4]] = dict(ref_case(p), enc_sha256=hashlib.sha256(data).hexdigest())
    with tempfile.TemporaryDirectory() as td:
        for name, (raw, level) in sorted(datagen.corpus_small().items()):
            enc = datagen.compress(raw, level)
            p = os.path.join(td, "x.bz2")
            open(p, "wb").write(enc)
            golden["generated"][name] = dict(ref_case(p), enc_sha256=hashlib.sha256(enc).hexdigest(),
                                             raw_sha256=hashlib.sha256(raw).hexdigest())
        parts = [datagen.text_like(250_000, 31), datagen.random_bytes(150_000, 32), b"", b"x",
                 datagen.runs(99_999, 33)]
        enc = datagen.multistream(parts, 1)
        p = os.path.join(td, "m.bz2")
        open(p, "wb").write(enc)
        golden["multistream"]["five-streams-l1"] = dict(ref_case(p), enc_sha256=hashlib.sha256(enc).hexdigest())
        garbage = enc + b"\x00" * 37 + datagen.compress(b"must not be read", 9)
        open(p, "wb").write(garbage)
        golden["multistream"]["trailing-garbage"] = dict(ref_case(p), enc_sha256=hashlib.sha256(garbage).hexdigest())
        for name, data in corrupt_cases().items():
            open(p, "wb").write(data)
            out = O.ref_run("probe", p, 32).strip()
            verdict = out.split()[0:2] if out.startswith("EXC") else ["OK", ""]
            golden["probes"][name] = {"enc_sha256": hashlib.sha256(data).hexdigest(), "verdict": verdict[0],
                                      "exception": verdict[1] if verdict[0] == "EXC" else "",
                                      "what": " ".join(out.split()[2:])[:160] if verdict[0] == "EXC" else ""}
        # damaged corpus: what the reference does with every block offset of every damaged file
        golden["damaged"] = {}
        for name, (data, offs) in sorted(datagen.damaged_corpus().items()):
            open(p, "wb").write(data)
            entry = {"enc_sha256": hashlib.sha256(data).hexdigest(), "blocks": {}}
            for off in offs:
                out = O.ref_run("probe", p, off).strip()
                if out.startswith("EXC"):
                    parts = out.split(" ", 2)
                    entry["blocks"][str(off)] = {"verdict": "EXC", "exception": parts[1],
                                                 "what": (parts[2] if len(parts) > 2 else "")[:160]}
                else:
                    f = out.split()
                    entry["blocks"][str(off)] = {"verdict": "OK", "exception": "", "what": "", "size": int(f[2]),
                                                 "header_crc": int(f[3], 16), "calc_crc": int(f[4], 16), "decoded": int(f[5])}
            golden["damaged"][name] = entry
    with open(OUT, "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
