import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "avx-ecm_amd"))
import pyecm
N2 = 928633991635461948591207425996748469032239132452053954416571123490620601752297841488884065228486244831964878827474780904585157939329250430073544670210094362165291423713860183
ref = [8235548891707629320749, 6221122412999520276841, 6221122412999520276841, 21403655900251913, 6221122412999520276841,
       2393719934217064415480942093, 432598094069, 48380458277928419578409]
sig0 = 768295079280089151
for env in ({}, {"GECM_S2_SUBSEQ": "1"}, {"GECM_S2_SUBSEQ": "1", "GECM_S2_SLICES": "1"}, {"GECM_S2_SUBSEQ": "4"}):
    for k in ("GECM_S2_SUBSEQ", "GECM_S2_SLICES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for batch in (8, 9, 70):
        eng = pyecm.Engine(N2)
        eng.build_curves([sig0 + k for k in range(batch)])
        eng.stage1(65)
        eng.stage2(50085)
        st = eng.stage2_stats()
        got = [eng.stage2_factor(k) for k in range(8)]
        got = [g[0] if g else None for g in got]
        print(env, "batch", batch, "D", st.D, "U", st.U, [("ok" if g == r else "DIFF %s" % g) for g, r in zip(got, ref)], flush=True)
        eng.close()
