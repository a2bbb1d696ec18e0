import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from openpvsg_amd import ops
dev='cuda:0'; g=torch.Generator().manual_seed(0)
for shapes in ([(23,40),(46,80),(92,160)], [(5,7)], [(4,4),(8,8),(16,16),(32,32)]):
    B=32 if len(shapes)==3 else 3; M,D,P,L=8,32,4,len(shapes)
    S=sum(h*w for h,w in shapes)
    v=torch.randn(B,S,M,D,generator=g).to(dev)
    loc=(torch.rand(B,S,M,L,P,2,generator=g)*1.4-0.2).to(dev)
    w=torch.softmax(torch.randn(B,S,M,L*P,generator=g),-1).view(B,S,M,L,P).to(dev)
    ss=torch.tensor(shapes,dtype=torch.long,device=dev); lsi=torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
    outs={}
    for c in ('0','1'):
        os.environ['PVSG_MSDA_COOP']=c
        f=lambda: ops.ms_deform_attn_forward(v,ss,lsi,loc,w)
        o=f(); torch.cuda.synchronize()
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize()
        outs[c]=(o, s.elapsed_time(e)/20)
    print(shapes[:2], 'B',B,'coop0 %.3f ms coop1 %.3f ms'%(outs['0'][1],outs['1'][1]), 'equal', torch.equal(outs['0'][0],outs['1'][0]), float((outs['0'][0]-outs['1'][0]).abs().max()))
