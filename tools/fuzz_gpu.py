"""Damaged-index fuzz on the GPU box: python tools/fuzz_gpu.py [seed] [trials] [only_trial]
Mutates block files of the committed fixtures (random bytes, extreme header words, truncation) and, when the loader
still accepts the index, runs count / locate / leaf requests in a child process under a time limit.  Expected: an
error code or (garbage) results -- never a hang, and no GPU fault."""
import os, shutil, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[2])
import femto_amd
try:
    ix = femto_amd.Index(sys.argv[1], device=0)
except femto_amd.FemtoAmdError as e:
    print("ERR_OPEN", e.code); sys.exit(0)
n = ix.info.total_length
rng = np.random.Generator(np.random.PCG64(1))
pats = [rng.integers(5, 261, int(rng.integers(0, 12))).astype(np.uint16) for _ in range(300)]
def stage(*a):
    print("stage", *a, file=sys.stderr, flush=True)
try:
    for mode in (ix.rank_mode, 1, 0):
        ix.set_rank_mode(mode)
        stage(mode, "count"); ix.count(pats)
        stage(mode, "locate"); ix.locate(pats, 5)
        stage(mode, "block_requests"); ix.block_requests(rng.integers(0, max(1, n), 500).astype(np.int64))
        stage(mode, "locate_all"); ix.locate([np.zeros(0, dtype=np.uint16)], min(n, 20000))
    print("RAN")
except femto_amd.FemtoAmdError as e:
    print("ERR_RUN", e.code)
'''
def main():
    from conftest import Fixture
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1      # reproduce one trial (the mutations depend on the seed alone)
    keep = os.environ.get("FUZZ_KEEP")                          # directory that receives the mutated index of a bad trial
    names = ["acgt48k", "eng2doc", "runs3doc", "chunks2doc", "b1000", "bytes256", "counter400_small"]
    root = tempfile.mkdtemp()
    open(root + "/child.py", "w").write(CHILD)
    rng = np.random.Generator(np.random.PCG64(seed))
    stats, bad = {}, 0
    for t in range(trials):
        name = names[t % len(names)]
        if not os.path.exists(os.path.join(root, name)):
            Fixture(name, root)
        dst = os.path.join(root, "mut")
        shutil.rmtree(dst, ignore_errors=True)
        shutil.copytree(os.path.join(root, name, "index"), dst)
        files = sorted(f for f in os.listdir(dst) if f != "_femto_index")
        f = files[int(rng.integers(0, len(files)))]
        data = bytearray(open(os.path.join(dst, f), "rb").read())
        kind = int(rng.integers(0, 3))
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))
        elif kind == 1:
            pos = int(rng.integers(0, min(len(data) - 4, 4096))) & ~3
            data[pos:pos + 4] = int(rng.choice([0, 0xffffffff, 0x7fffffff, 0x80000000, len(data)])).to_bytes(4, "big")
        else:
            pos = int(rng.integers(0, max(1, len(data) - 4))) & ~3
            data[pos:pos + 4] = int(rng.integers(0, 2**32)).to_bytes(4, "big")
        open(os.path.join(dst, f), "wb").write(data)
        if only >= 0 and t != only:
            continue
        if only >= 0 and keep and not os.environ.get("FUZZ_RUN"):
            shutil.copytree(dst, os.path.join(keep, f"fuzz_s{seed}_t{t}"), dirs_exist_ok=True)
            print("kept", name, f, "kind", kind)
            continue
        try:
            r = subprocess.run([sys.executable, root + "/child.py", dst, ROOT], capture_output=True, text=True, timeout=60)
            out = r.stdout.strip().split()[0] if r.stdout.strip() else f"DIED_rc{r.returncode}"
        except subprocess.TimeoutExpired:
            out = "TIMEOUT"
        stats[out] = stats.get(out, 0) + 1
        if out.startswith("DIED") or out == "TIMEOUT":
            bad += 1
            stages = [l for l in (r.stderr.splitlines() if out != "TIMEOUT" else []) if l.startswith("stage")]
            print("BAD trial", t, out, name, f, "kind", kind, "last", stages[-1:] , (r.stderr[-300:].replace("\n", " | ") if out != "TIMEOUT" else ""), flush=True)
            if keep:
                shutil.copytree(dst, os.path.join(keep, f"fuzz_s{seed}_t{t}"), dirs_exist_ok=True)
    print("stats", stats, "bad", bad)
if __name__ == "__main__":
    main()
