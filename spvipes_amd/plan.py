"""Transport plans stored sparse on device.

The reference keeps the optimal-transport plan as a dense float32 tensor on the model
(/root/reference/src/spVIPES/model/spvipes.py:245) and gathers the [B0, B1] block of every minibatch pair from it
(module/spVIPESmodule.py:474-482): 500 k x 500 k cells would need 1 TB.  A plan produced by entropic / k-NN-restricted OT
has a handful of entries per row, so it lives here as CSR (rows = cells of group 0) plus the CSR of its transpose (rows =
cells of group 1), int32 indices, fp32 values; the PoE kernels walk only the stored entries of the minibatch's rows."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _abi
from ._abi import SpvPlan, ptr, stream_ptr


class SparsePlan:
    def __init__(self, indptr0, indices0, values0, indptr1, indices1, values1, n0: int, n1: int, device):
        dev = torch.device(device)
        i32 = lambda t: torch.as_tensor(np.asarray(t) if not isinstance(t, torch.Tensor) else t).to(dev, torch.int32).contiguous()
        f32 = lambda t: torch.as_tensor(np.asarray(t) if not isinstance(t, torch.Tensor) else t).to(dev, torch.float32).contiguous()
        self.ptr0, self.ind0, self.val0 = i32(indptr0), i32(indices0), f32(values0)
        self.ptr1, self.ind1, self.val1 = i32(indptr1), i32(indices1), f32(values1)
        self.n0, self.n1, self.device = int(n0), int(n1), dev
        if self.ptr0.numel() != self.n0 + 1 or self.ptr1.numel() != self.n1 + 1:
            raise ValueError("SparsePlan: indptr length must be n_rows + 1")
        if self.ind0.numel() >= 2 ** 31 or self.ind1.numel() >= 2 ** 31:
            raise _abi.SpvError("SparsePlan: more than 2^31 stored entries are not supported (int32 indptr)")
        self.inv0 = torch.empty(self.n0, dtype=torch.int32, device=dev)
        self.inv1 = torch.empty(self.n1, dtype=torch.int32, device=dev)

    # ---- constructors ------------------------------------------------------------------------------------------
    @classmethod
    def from_dense(cls, plan: torch.Tensor, device=None) -> "SparsePlan":
        """From the reference's dense [n0, n1] tensor (entries <= 0 are dropped: a transport plan is non-negative)."""
        plan = plan.detach()
        dev = torch.device(device) if device is not None else plan.device
        p = plan.to(dev, torch.float32)
        p = torch.where(p > 0, p, torch.zeros_like(p))
        a, b = p.to_sparse_csr(), p.t().contiguous().to_sparse_csr()
        return cls(a.crow_indices(), a.col_indices(), a.values(), b.crow_indices(), b.col_indices(), b.values(), p.shape[0], p.shape[1], dev)

    @classmethod
    def from_scipy(cls, mat, device) -> "SparsePlan":
        """From any scipy.sparse matrix [n0, n1] (never densified)."""
        a = mat.tocsr()
        a.eliminate_zeros()
        b = a.T.tocsr()
        return cls(a.indptr, a.indices, a.data, b.indptr, b.indices, b.data, a.shape[0], a.shape[1], device)

    @classmethod
    def from_any(cls, plan, device) -> "SparsePlan":
        if isinstance(plan, SparsePlan):
            return plan
        if isinstance(plan, torch.Tensor):
            return cls.from_dense(plan, device)
        if hasattr(plan, "tocsr"):
            return cls.from_scipy(plan, device)
        return cls.from_dense(torch.as_tensor(np.asarray(plan)), device)

    # ---- per-step helpers ----------------------------------------------------------------------------------------
    def c_struct(self) -> SpvPlan:
        return SpvPlan(ptr0=ptr(self.ptr0), ind0=ptr(self.ind0), val0=ptr(self.val0), n0=self.n0,
                       ptr1=ptr(self.ptr1), ind1=ptr(self.ind1), val1=ptr(self.val1), n1=self.n1)

    def bind_minibatch(self, idx0: torch.Tensor, idx1: torch.Tensor) -> None:
        """inv0 / inv1 <- position of every dataset cell in the minibatch (or -1)."""
        _abi.call("spv_plan_invmap", ptr(idx0), idx0.numel(), ptr(idx1), idx1.numel(), ptr(self.inv0), self.n0, ptr(self.inv1), self.n1,
                  stream_ptr())

    def dense_block(self, idx0: torch.Tensor, idx1: torch.Tensor) -> torch.Tensor:
        """plan[idx0][:, idx1] as a dense tensor (torch ops; for tests and for the torch PoE paths)."""
        a = torch.sparse_csr_tensor(self.ptr0.long(), self.ind0.long(), self.val0, size=(self.n0, self.n1)).to_dense()
        return a[idx0.long()][:, idx1.long()]
