"""Generate tests/golden/chunk_vectors.json from the REAL rapidgzip::Bzip2Chunk<ChunkData>::decodeChunk
(oracle/_ref/ref_chunk, compiled from /root/reference/src/rapidgzip/chunkdecoding/Bzip2Chunk.hpp by oracle/Makefile).
Authoring container only; the JSON holds the reference's answers (data), the inputs are reproducible from
tests/chunk_cases.py + CPython bz2 and pinned by their sha256.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import chunk_cases

REF = os.path.join(ROOT, "oracle", "_ref", "ref_chunk")
OUT = os.path.join(ROOT, "tests", "golden", "chunk_vectors.json")


def main():
    files = chunk_cases.inputs()
    out = {"inputs": {name: hashlib.sha256(data).hexdigest() for name, data in files.items()}, "cases": []}
    with tempfile.TemporaryDirectory() as tmp:
        paths = {}
        for name, data in files.items():
            paths[name] = os.path.join(tmp, name + ".bz2")
            with open(paths[name], "wb") as f:
                f.write(data)
        for name, start, until, limit in chunk_cases.requests(files):
            res = subprocess.run([REF, paths[name], str(start), str(until), str(limit)], capture_output=True, text=True,
                                 check=True).stdout
            out["cases"].append({"input": name, "start": start, "until": until, "max_decoded": limit,
                                 "reference": json.loads(res)})
    with open(OUT, "w") as f:
        json.dump(out, f, indent=0)
        f.write("\n")
    print(len(out["cases"]), "cases ->", OUT)


if __name__ == "__main__":
    main()
