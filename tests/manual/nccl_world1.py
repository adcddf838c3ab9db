"""Manual check (1 GPU): RCCL process group of size 1, device-to-device export of the packed parameters into a torch
tensor, all_gather_into_tensor - the calls bench.py makes at N > 1.   python tests/manual/nccl_world1.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bodyfitting_amd import native as N, synthetic as S   # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
model, gmm = S.make_model("smpl"), S.make_gmm()
dev = N.DeviceModel(model, gmm, device=0)
c2w, K, kp, ndiv, betas, pose = N.pack_problem([S.make_problem(model, f, 8) for f in range(2)])
b = N.FrameBatch(dev, 2, 8)
b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
b.fit(10)
send = torch.empty(2 * dev.n_params, dtype=torch.float32, device="cuda:0")
recv = torch.empty(2 * dev.n_params, dtype=torch.float32, device="cuda:0")
b.export_params_dev(send.data_ptr())
b.sync()
dist.all_gather_into_tensor(recv, send)
torch.cuda.synchronize()
dist.barrier()
assert np.array_equal(recv.cpu().numpy().reshape(2, -1), b.get_params())
print("nccl world-1 gather ok")
dist.destroy_process_group()
