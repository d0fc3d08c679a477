"""-m gpu: the batch-aware CLI (apps/aligner_batch.cpp -> build/edlib-aligner-batch) prints what the
reference's CLI prints.  Expected output comes from oracle/_ref/aligner_ref = the reference's
aligner.cpp + the reference's edlib.cpp (pure CPU, prebuilt where /root/reference exists)."""
import os
import re
import subprocess

import pytest

from edlib_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _norm(text):
    out = []
    for line in text.replace("\r", "\n").split("\n"):
        if re.fullmatch(r"\d+/\d+", line.strip()) or "time of searching" in line:
            continue
        out.append(line.rstrip())
    return "\n".join(l for l in out if l != "")


def _fasta(path, seqs):
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">seq%d some description\n" % i)
            b = bytes(s)
            for j in range(0, len(b), 60):
                f.write(b[j:j + 60].decode() + "\n")


@pytest.mark.parametrize("flags", [
    ["-m", "HW"], ["-m", "HW", "-l"], ["-m", "HW", "-p", "-f", "CIG_EXT"], ["-m", "HW", "-p"],
    ["-m", "HW", "-p", "-f", "CIG_STD"], ["-m", "SHW", "-l"], ["-m", "NW"], ["-m", "HW", "-n", "5"],
    ["-m", "HW", "-n", "7", "-k", "3", "-l"], ["-m", "HW", "-k", "2"], ["-m", "HW", "-s"]])
def test_batch_cli_matches_reference_cli(tmp_path, flags):
    ref = os.path.join(ROOT, "oracle", "_ref", "aligner_ref")
    exe = os.path.join(ROOT, "build", "edlib-aligner-batch")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/aligner_ref was not prebuilt")
    assert os.path.exists(exe), "build/edlib-aligner-batch missing: run make"
    target = synth.random_dna(31, 20000)
    reads = synth.illumina_reads(target, 40, m=150, seed=32)["reads"]
    q, t = str(tmp_path / "q.fasta"), str(tmp_path / "t.fasta")
    _fasta(q, list(reads))
    _fasta(t, [target])
    want = subprocess.run([ref] + flags + [q, t], capture_output=True, text=True, timeout=300)
    got = subprocess.run([exe] + flags + [q, t], capture_output=True, text=True, timeout=300)
    assert got.returncode == 0, got.stdout[-500:] + got.stderr[-500:]
    assert _norm(got.stdout) == _norm(want.stdout)
