"""Test double for the CorrField device interface, backed by the CPU oracle -- lives in tests/ only.  It lets the
multi-process (gloo, CPU) tests drive correrender_amd.distributed.ShardedCorrField end to end without a GPU: what is
under test there is the sharding / ownership / broadcast / all-reduce logic, not the kernels."""
import numpy as np
import torch

import oracle_lib


class OracleEngine:
    def __init__(self):
        self.oracle = oracle_lib.load_oracle()
        self.members = None
        self.grid = None
        self.cs = 0

    def set_grid(self, xs, ys, zs, cs):
        self.grid, self.cs = (xs, ys, zs), cs

    def bind_members(self, members):
        arr = members.numpy() if isinstance(members, torch.Tensor) else np.stack([np.asarray(m) for m in members])
        xs, ys, zs = self.grid
        self.members = np.ascontiguousarray(arr, np.float32).reshape(self.cs, zs, ys, xs)

    def member_minmax(self):
        return self.oracle.minmax(self.members)

    def gather_reference_device(self, x, y, z, out, stream=0):
        out.copy_(torch.from_numpy(self.members[:, z, y, x].copy()))

    def gather_reference_rows_device(self, points, out, stream=0):
        for r, p in enumerate(points):
            if p is None:
                out[r].zero_()
            else:
                out[r].copy_(torch.from_numpy(self.members[:, p[2], p[1], p[0]].copy()))

    def prepare_device(self, measure, slot, ref=None, *, device_reference=None, stream=0, **kw):
        # the engine double "prepares" by remembering the reference vector of the slot
        self.prepared = getattr(self, "prepared", {})
        self.prepared[slot] = (int(measure), device_reference.numpy().copy() if device_reference is not None
                               else self.members[:, ref[2], ref[1], ref[0]].copy())

    def compute_device(self, measure, out, ref=None, *, device_reference=None, stream=0, k=None,
                       kraskov_estimator_index=1, num_bins=80, minmax_ref=None, minmax_query=None,
                       reference_values=None, prepared_slot=None):
        if prepared_slot is not None:
            pm, refv = self.prepared[prepared_slot]
            assert pm == int(measure)
        else:
            refv = (device_reference.numpy().copy() if device_reference is not None
                    else self.members[:, ref[2], ref[1], ref[0]].copy())
        kw = dict(k=k if k is not None else max(-(-3 * self.cs // 100), 1), estimator=kraskov_estimator_index,
                  num_bins=num_bins)
        if minmax_ref is not None:
            kw.update(minmax_ref=minmax_ref, minmax_query=minmax_query)
        res = self.oracle.field(int(measure), self.members, refv, **kw)
        out.copy_(torch.from_numpy(res))
        return out
