import sys; sys.path.insert(0,'.')
import numpy as np, torch
from ofdm_amd.batch import TxContext
from oracle import c_oracle
ctx=TxContext(0)
for ln in (0,100):
    rng=np.random.default_rng(1)
    n=3
    pay=rng.integers(0,256,n*ln,dtype=np.uint8)
    plan=ctx.plan_uniform(n,8,ln)
    t=plan.encode_bits(torch.from_numpy(pay).cuda(),taps=True)
    iq=plan.modulate(t["interleaved"]).cpu().numpy()
    iq2=plan.tx(torch.from_numpy(pay).cuda()).cpu().numpy()
    lay=plan.layout()
    for i in range(n):
        ref=c_oracle.encode_bits(pay[i*ln:(i+1)*ln].tobytes(),8)
        mo,ml=int(lay["msg_off"][i]),int(lay["msg_len"][i])
        got=t["interleaved"].cpu().numpy()[mo:mo+ml].tobytes()
        cod=t["coded"].cpu().numpy()[mo:mo+ml].tobytes()
        print(ln,i,"itl ok",got==ref["interleaved"],"cod ok",cod==ref["coded"])
        want=c_oracle.modulate(ref["interleaved"],8,int(lay["pilot_base"][i]),0.3,3)["iq64"]
        a=int(lay["sample_off"][i])
        for name,x in (("2k",iq),("tx",iq2)):
            e=np.abs(x[a:a+len(want)]-want).reshape(-1,80).max(axis=1)
            print("   ",name,"bad rows",np.nonzero(e>1e-4)[0].tolist())
