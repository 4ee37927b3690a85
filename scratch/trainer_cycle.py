import sys; sys.path.insert(0,'/root/repo')
import torch
from glfusion_amd.engine import Trainer
cfg={"train":{"batch_size":2,"num_epochs":1,"clip_length":40,"view_num":["1","3"],"test_view":["1","3"],"dense_cyc":False,
              "save_dir":"/tmp/glf_ckpt","iters_per_epoch":2,"global_rank":0},
     "net":{"opt":{"opt_name":"Adam","lr":3e-4,"params":(0.9,0.999),"weight_decay":1e-5}}}
t=Trainer(cfg)
t.train(is_backbone=False,is_cycle=True)
cfg["train"]["dense_cyc"]=True
t2=Trainer(cfg); t2.train(is_cycle=True)
print(t2.eval())
