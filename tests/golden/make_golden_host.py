"""Generate tests/golden/host_vectors.txt: scripts for the host-side scheduler classes and the answers of the REAL
reference classes (oracle/_ref/ref_host = FetchNextAdaptive of src/core/Prefetcher.hpp:82-217 and BlockMap of
src/core/BlockMap.hpp:26-295 behind oracle/ref_host_harness.cpp).  Authoring container only.  Each line of the output is
"<command> => <answer>"; tests/native/host_known_answers.cpp replays the commands on this repository's own classes.

The scripts: the access sequences of src/tests/core/testPrefetcher.cpp:239-285 (linear, duplicate, one random seek,
many random seeks), sequential and backward sweeps (:1075-1100), and seeded random mixes; block-map scripts with
end-of-stream entries, duplicate / inconsistent / non-increasing pushes, look-ups, finalize, export and import.
"""
import os
import random
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_host")
OUT = os.path.join(ROOT, "tests", "golden", "host_vectors.txt")


def strategy_scripts():
    cmds = []
    # testLinearAccess<FetchNextAdaptive>, testPrefetcher.cpp:239-285
    cmds += ["strategy 3", "p 4", "f 23", "p 3", "p 3", "f 23", "p 3"]
    for index in range(24, 40):
        cmds += [f"f {index}", "p 8", "s"]
    cmds += ["f 3"] + [f"p {n}" for n in range(1, 10)] + ["s"]
    for i in range(0, 600, 10):
        cmds += [f"f {i}"]
    cmds += ["p 10", "s"]
    # testSplit's first half, :306-313
    cmds += ["strategy 3", "p 4", "f 0", "p 4"]
    # sequential and backward sweeps, :1075-1100
    cmds += ["strategy 3"]
    for i in range(0, 60):
        cmds += [f"f {i}", "p 16"]
    cmds += ["strategy 3"]
    for i in range(60, 0, -1):
        cmds += [f"f {i}", "p 16", "s"]
    # seeded random mixes for several memory sizes and prefetch limits
    r = random.Random(0xFE7C4)
    for memory in (3, 1, 2, 5, 8):
        cmds.append(f"strategy {memory}")
        index = 0
        for _ in range(700):
            roll = r.random()
            if roll < 0.55:
                index += 1
            elif roll < 0.65:
                pass                      # duplicate access
            elif roll < 0.75:
                index += r.choice([2, 3, -1, -2])
                index = max(0, index)
            else:
                index = r.randrange(0, 5000)
            cmds.append(f"f {index}")
            if r.random() < 0.8:
                cmds.append(f"p {r.choice([0, 1, 2, 3, 4, 7, 8, 16, 24, 64, 100, 512, 1536, 2560, 10240])}")
            if r.random() < 0.3:
                cmds.append("s")
    return cmds


def map_scripts():
    cmds = []
    r = random.Random(0xB10C)
    for case in range(40):
        cmds.append("map")
        cmds.append("state")
        cmds.append("find 0")
        enc, dec = (0 if case % 7 == 0 else 32), 0
        history = []
        for _ in range(r.randrange(0, 40)):
            enc_size = r.randrange(80, 8_000_000)
            dec_size = 0 if r.random() < 0.2 else r.randrange(1, 1_000_000)
            cmds.append(f"push {enc} {enc_size} {dec_size}")
            history.append((enc, enc_size, dec_size))
            roll = r.random()
            if roll < 0.15 and history:       # the same block again (prefetched twice): consistent or not
                e, es, ds = r.choice(history)
                cmds.append(f"push {e} {es} {ds if r.random() < 0.6 else ds + 1}")
            elif roll < 0.22:                 # an offset that was never pushed, below the last one
                cmds.append(f"push {max(0, enc - r.randrange(1, 50))} 100 100")
            enc += enc_size
            dec += dec_size
            for _ in range(r.randrange(0, 4)):
                cmds.append(f"find {r.randrange(0, dec + 10)}")
            if r.random() < 0.1:
                cmds.append("state")
            if r.random() < 0.05:
                cmds.append("dump")
        cmds += ["state", f"find {dec}", f"find {dec + 5}", "finalize", "finalize", "state", "dump",
                 f"find {max(0, dec - 1)}", f"find {dec}", f"find {dec + 1000}", f"push {enc + 10} 5 5"]
    # import: with end-of-stream entries (equal decoded offsets), single entries, maps that start at 0
    cmds += ["map", "set 32:0 211:1 296:1", "state", "dump", "find 0", "find 1", "find 2"]
    cmds += ["map", "set 0:0", "state", "dump", "find 0"]
    cmds += ["map", "set 32:0 172368:108614 358483:216269 423736:250000 423848:250000 777057:356231 870528:400000 870608:400000",
             "state", "dump"] + [f"find {x}" for x in (0, 108613, 108614, 249999, 250000, 250001, 399999, 400000, 400001)]
    cmds += ["map", "push 32 100 10", "set 32:0 500:100 600:100", "state", "dump", "push 700 10 10"]
    return cmds


def main():
    cmds = strategy_scripts() + map_scripts()
    out = subprocess.run([REF], input="\n".join(cmds) + "\n", capture_output=True, text=True, check=True).stdout.splitlines()
    assert len(out) == len(cmds), (len(out), len(cmds))
    assert "NOT-CONSECUTIVE" not in out and "?" not in out
    with open(OUT, "w") as f:
        for c, a in zip(cmds, out):
            f.write(f"{c} => {a}\n")
    print(len(cmds), "commands ->", OUT)


if __name__ == "__main__":
    main()
