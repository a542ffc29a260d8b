import sys, random, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import gen_golden as G
from oracle import deepsee_oracle as O
G.install_torchvision_stub(); sys.path.insert(0,'/root/reference')
from managers.trainer_manager import TrainerManager
torch.autograd.set_detect_anomaly(False)
name=sys.argv[1]; spec=G.CASES[name]
opt=O.make_opt(**spec['opt']); states=O.recipe_state(opt)
batch=O.synthetic_batch(opt, spec['n'], seed=1000+spec['seed'])
res=[]
for pert in (0.0, 1e-7):
    tm=TrainerManager(G.ref_namespace(opt)); m=tm.sr_model_on_one_gpu
    for net,mod in (('SR',m.netSR),('D',m.netD),('E',m.netE)): mod.load_state_dict(states[net])
    vgg=m.criterionVGG.vgg
    for sl in (vgg.slice1,vgg.slice2,vgg.slice3,vgg.slice4,vgg.slice5):
        for idx,mod in sl.named_children():
            if hasattr(mod,'weight'):
                mod.weight.data.copy_(states['VGG']['features.%s.weight'%idx]); mod.bias.data.copy_(states['VGG']['features.%s.bias'%idx])
    torch.manual_seed(1); random.seed(1)
    b={k:v.clone() for k,v in batch.items()}; b['image']=b['image']*(1+pert)
    tm.run_generator_one_step(b)
    g={k:p.grad.clone() for k,p in m.netSR.named_parameters() if p.grad is not None}
    fake=tm.generated.detach().clone()
    tm.run_discriminator_one_step(b)
    d={k:p.grad.clone() for k,p in m.netD.named_parameters() if p.grad is not None}
    res.append((fake,g,d))
(f0,g0,d0),(f1,g1,d1)=res
print('fake rel', float((f0-f1).norm()/f0.norm()))
gm=max(float(v.norm()) for v in g0.values())
es=sorted(((float((g0[k]-g1[k]).norm())/max(float(g0[k].norm()),1e-3*gm),k) for k in g0), reverse=True)[:5]
print('G grads worst', es)
es=sorted(((float((d0[k]-d1[k]).norm())/float(d0[k].norm()),k) for k in d0), reverse=True)[:5]
print('D grads worst', es)
