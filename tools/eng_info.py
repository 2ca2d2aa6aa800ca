import sys, os
sys.path.insert(0, "/root/repo")
import femto_amd
path = "/tmp/femto_amd_bench/eng_2p30_s20260928"
import glob
c = [p for p in glob.glob("/tmp/femto_amd_bench/*") if "eng" in p]
print(c)
path = c[0]
for b in (-1, 32 << 30, 64 << 30):
    ix = femto_amd.Index(path, device=0, options={"hbm_budget_bytes": b})
    pi, st = ix.pack_info(), ix.structures()
    print(b >> 30, {k: pi[k] for k in ("sa_full", "isa_full", "char_rank_lines", "context_table", "context_syms", "context2_syms", "ktab_syms")}, {k: st[k] >> 20 for k in ("context_tables", "char_rank_lines", "text_sa_isa", "two_level_lines", "level_table", "hbm_allocated")})
    ix.close()
