import sys, tempfile, os
sys.path.insert(0,'.')
import harc_amd
from tests import oracle_lib as ol
case = sys.argv[1]
g = ol.load_golden(case)
L = len(g["reads.txt"].split(b"\n")[0])
d = tempfile.mkdtemp()
base = ol.stage_dir(d, {k[len("stage1/"):]: v for k, v in g.items() if k.startswith("stage1/")})
harc_amd.encoder(base, L, num_thr=1)
got = ol.read_dir(base)
for f in ol.stage2_files(1):
    a, b = got[f], g["stage2/"+f]
    if a != b:
        n = min(len(a), len(b)); first = next((i for i in range(n) if a[i]!=b[i]), n)
        print(f, len(a), len(b), "first diff", first, a[first:first+12], b[first:first+12])
    else: print(f, "ok")
