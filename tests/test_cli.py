"""Command line tool `ibzip2-mi355x` against the behaviour of the reference `ibzip2` (src/tools/ibzip2.cpp:172-480):
option surface, output-file rules, refusal to overwrite, the two offset list formats (ibzip2.cpp:68-93) and -t.
The reference binary itself cannot be built here (cxxopts is an empty submodule), so expectations come from reading
its source and from the oracle's block map."""
import os
import subprocess

import pytest

from conftest import FIXTURES, ROOT, fixture_names, read_fixture
import datagen

CLI = os.path.join(ROOT, "indexed_bzip2_amd", "ibzip2-mi355x")


@pytest.fixture(scope="module")
def cli(native):
    assert os.path.exists(CLI), "python -m indexed_bzip2_amd.build builds the tool next to the library"
    return CLI


def run(cli, *args, stdin=None, cwd=None):
    return subprocess.run([cli, *args], input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=cwd, timeout=300)


# ------------------------------------------------------------------------------------------------ CPU only

def test_help_version_and_no_arguments(cli):
    r = run(cli, "--help")
    assert r.returncode == 0
    for opt in (b"-c, --stdout", b"-d, --decompress", b"-f, --force", b"-i, --input", b"-o, --output", b"-k, --keep",
                b"-t, --test", b"-p, --block-finder-parallelism", b"-P, --decoder-parallelism", b"-q, --quiet",
                b"-v, --verbose", b"-V, --version", b"-l, --list-compressed-offsets", b"-L, --list-offsets",
                b"--buffer-size"):
        assert opt in r.stdout, opt
    r = run(cli, "-V")
    assert r.returncode == 0 and b"ibzip2-mi355x" in r.stdout
    # a file but nothing to do with it: help + exit code 1 (ibzip2.cpp:475-479)
    r = run(cli, "-q", os.path.join(FIXTURES, "1B.bz2"), "-c")
    assert r.returncode == 1 and b"No suitable arguments" in r.stderr
    r = run(cli, "--no-such-option")
    assert r.returncode == 1
    r = run(cli, "-d", "a.bz2", "b.bz2")
    assert r.returncode == 1 and b"One or none bzip2 filename" in r.stderr


@pytest.mark.parametrize("name", fixture_names())
def test_list_compressed_offsets_without_decoding(cli, oracle, name, tmp_path):
    """-l alone uses the block finder only (ibzip2.cpp:461-473): block AND end-of-stream magics, sorted, one per line."""
    enc, raw = read_fixture(name)
    want = sorted(oracle.find_magic(enc, oracle.MAGIC_BLOCK) + oracle.find_magic(enc, oracle.MAGIC_EOS))
    path = os.path.join(FIXTURES, name + ".bz2")
    r = run(cli, "-l", "--", path)
    assert r.returncode == 0, r.stderr
    assert [int(x) for x in r.stdout.split()] == want
    out = tmp_path / "offsets.dat"
    r = run(cli, "-t", "-v", "-p", "2", "-l", str(out), "--", path)
    assert r.returncode == 0, r.stderr
    assert [int(x) for x in out.read_text().split()] == want
    assert b"Found %d blocks" % len(want) in r.stdout
    # existing list file is not overwritten without --force (ibzip2.cpp:351-356)
    r = run(cli, "-l", str(out), "--", path)
    assert r.returncode == 1 and b"already exists" in r.stderr
    r = run(cli, "-f", "-l", str(out), "--", path)
    assert r.returncode == 0


def test_list_compressed_offsets_from_stdin(cli, oracle):
    enc = datagen.multistream([datagen.text_like(30_000, 3), b"x" * 1000], 1)
    want = sorted(oracle.find_magic(enc, oracle.MAGIC_BLOCK) + oracle.find_magic(enc, oracle.MAGIC_EOS))
    r = run(cli, "-l", stdin=enc)
    assert r.returncode == 0, r.stderr
    assert [int(x) for x in r.stdout.split()] == want


# ------------------------------------------------------------------------------------------------ GPU

@pytest.mark.gpu
def test_decompress_to_deduced_file_and_overwrite_rules(cli, tmp_path):
    raw = datagen.random_text_file(1_500_000, 5)
    enc = datagen.compress(raw, 9)
    src = tmp_path / "data.BZ2"            # suffix match is case-insensitive (ibzip2.cpp:319)
    src.write_bytes(enc)
    r = run(cli, "-d", "-P", "0", str(src))
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "data").read_bytes() == raw
    assert src.exists()                     # the tool never deletes anything
    r = run(cli, "-d", str(src))
    assert r.returncode == 1 and b"already exists" in r.stderr
    r = run(cli, "-d", "-f", str(src))
    assert r.returncode == 0 and (tmp_path / "data").read_bytes() == raw
    other = tmp_path / "noext"
    other.write_bytes(enc)
    r = run(cli, "-d", str(other))
    assert r.returncode == 0 and b"Could not deduce output file name" in r.stderr
    assert (tmp_path / "noext.out").read_bytes() == raw
    r = run(cli, "-d", "-o", str(tmp_path / "explicit.bin"), "--buffer-size", "70000", "-i", str(src))
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "explicit.bin").read_bytes() == raw


@pytest.mark.gpu
@pytest.mark.parametrize("name", fixture_names())
def test_stdout_test_and_offset_lists(cli, oracle, name, tmp_path):
    enc, raw = read_fixture(name)
    st, out, want_map, tg = oracle.decode_file(enc)
    path = os.path.join(FIXTURES, name + ".bz2")
    # decoded data on stdout -> both lists go to stderr (ibzip2.cpp:436-457), -L first
    r = run(cli, "-d", "-c", "-t", "-L", "-l", "-P", "3", path)
    assert r.returncode == 0, r.stderr
    assert r.stdout == raw
    lines = r.stderr.decode().split()
    pairs = [tuple(int(v) for v in ln.split(",")) for ln in lines if "," in ln]
    singles = [int(ln) for ln in lines if "," not in ln]
    assert dict(pairs) == want_map and [p[0] for p in pairs] == sorted(want_map)
    assert singles == sorted(want_map)
    # -L into a file, decoded data into a file: nothing on stdout
    lst = tmp_path / "map.csv"
    dst = tmp_path / "out.bin"
    r = run(cli, "-d", "-o", str(dst), "-L", str(lst), "--", path)
    assert r.returncode == 0, r.stderr
    assert dst.read_bytes() == raw and r.stdout == b""
    assert lst.read_text() == "".join(f"{k},{v}\n" for k, v in sorted(want_map.items()))
    # -L without -d decodes (to nowhere) and prints the map on stdout (outputFilePath deduced, ibzip2.cpp:440-445)
    r = run(cli, "-L", "--", path)
    assert r.returncode == 0, r.stderr
    assert r.stdout.decode() == "".join(f"{k},{v}\n" for k, v in sorted(want_map.items()))


@pytest.mark.gpu
def test_stdin_to_stdout_and_corrupt_input(cli):
    parts = [datagen.text_like(400_000, 11), datagen.random_bytes(100_000, 12)]
    enc = datagen.multistream(parts, 2)
    r = run(cli, "-d", "-P", "8", stdin=enc)
    assert r.returncode == 0, r.stderr
    assert r.stdout == b"".join(parts)
    bad = bytearray(enc)
    bad[len(bad) // 3] ^= 0x10
    r = run(cli, "-d", "-t", stdin=bytes(bad))
    assert r.returncode == 1 and b"Decoding failed" in r.stderr
    r = run(cli, "-d", stdin=b"not a bzip2 file at all")
    assert r.returncode == 1


@pytest.mark.gpu
def test_test_mode_checks_the_stream_crc(cli, oracle):
    """-t verifies the combined CRC in the end-of-stream block (what the reference's default serial decoder does,
    BZ2Reader.hpp:406-416); plain -d with the parallel reader semantics does not."""
    enc = datagen.compress(datagen.text_like(300_000, 71), 9)
    eos = oracle.find_magic(enc, oracle.MAGIC_EOS)[0]
    bad = bytearray(enc)
    bit = eos + 48 + 9
    bad[bit >> 3] ^= 0x80 >> (bit & 7)
    r = run(cli, "-d", "-t", stdin=bytes(bad))
    assert r.returncode == 1 and b"Stream CRC" in r.stderr
    r = run(cli, "-d", stdin=bytes(bad))
    assert r.returncode == 0 and len(r.stdout) == 300_000
    r = run(cli, "-d", "-t", stdin=enc)
    assert r.returncode == 0 and len(r.stdout) == 300_000
