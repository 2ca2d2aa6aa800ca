import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, tempfile
import femto_amd
from femto_amd import textgen as tg
from oracle import pyoracle as po
text = tg.t_acgt(1 << 22, 2024)
d = tempfile.mkdtemp()
path = d + "/ix"
femto_amd.build_index(path, [text], params=None, infos=["x"], device=0)
ix = femto_amd.Index(path, device=0)
o = po.Oracle(path)
plen_r, flat_r = tg.p_rand(20, 50000, 7)
plen_h, flat_h = tg.p_hit(20, 20, 50000, 8, text)
plen = np.concatenate([plen_r, plen_h]); flat = np.concatenate([flat_r, flat_h]); starts = tg.starts_of(plen)
of, ol = o.count_flat(plen, flat, starts, threads=8)
for n in [4096, 5000, 20000, 100000]:
    f, l = ix.count_flat(plen[:n], flat[:n*20], starts[:n])
    bad = np.nonzero((f != of[:n]) | (l != ol[:n]))[0]
    print(n, "bad", len(bad), bad[:10], [(int(f[i]), int(l[i]), int(of[i]), int(ol[i])) for i in bad[:5]])
    if len(bad):
        zero = np.sum((f[bad] == 0))
        print("  first==0:", zero, " last==total-1:", np.sum(l[bad] == ix.info.total_length - 1))
