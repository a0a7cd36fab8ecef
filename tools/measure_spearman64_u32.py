import os, sys
sys.path.insert(0, "/root/repo")
import torch
import correrender_amd as ca
xs = ys = zs = 256; n = xs*ys*zs
stream = torch.cuda.current_stream().cuda_stream
for cs in (33, 40, 48, 50, 56, 64):
  eng = ca.CorrField(0); eng.set_grid(xs, ys, zs, cs)
  block = torch.empty(cs*n, dtype=torch.float32, device="cuda")
  members = [block[c*n:(c+1)*n] for c in range(cs)]
  for c in range(cs): eng.synth_box_member(members[c], xs, ys, zs, 0, zs, c, cs, 1234, stream)
  torch.cuda.synchronize(); eng.bind_members(members); eng.set_profiling(True)
  outs = {}
  for rnd in range(2):
      for mode, w in (("0", "0"), ("1", "2")):
          os.environ["CRF_RANK_U32"] = mode; os.environ["CRF_RANK_U32_WAVES"] = w
          out = torch.empty(n, dtype=torch.float32, device="cuda")
          eng.compute_device(ca.Measure.SPEARMAN, out, (1, 2, 3), stream=stream); torch.cuda.synchronize(); eng.take_kernel_time()
          for i in range(5): eng.compute_device(ca.Measure.SPEARMAN, out, (17*i+5, 29, 31), stream=stream)
          torch.cuda.synchronize(); ms, cnt = eng.take_kernel_time()
          outs[(mode, w)] = out
          print(f"256^3 x {cs} Spearman u32={mode} waves={w}: {ms/cnt:.3f} ms  {eng.last_kernel_name()}  identical to shipped: {bool(torch.equal(out.view(torch.int32), outs[('0','0')].view(torch.int32)))}", flush=True)
