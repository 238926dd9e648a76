"""CSR containers of the mapping path: ``CSRData`` / ``CSRBatch`` with the
interface of ``torch_points3d/core/multimodal/csr.py`` (:44-303 and :305-479).

A CSRData is a list of variable-length groups stored flat: ``values[k]`` holds
one entry per item (a tensor, or a nested CSRData) and ``pointers`` (int64,
``pointers[0] == 0``) delimits the groups.  Only integer bookkeeping lives
here; it works on whatever device the tensors are on.
"""
import copy

import numpy as np
import torch


def tensor_idx(idx):
    """int / slice / list / ndarray / bool mask -> LongTensor
    (utils/multimodal.py:13-33)."""
    if idx is None:
        idx = torch.empty(0, dtype=torch.int64)
    elif isinstance(idx, (int, np.integer)):
        idx = torch.tensor([int(idx)], dtype=torch.int64)
    elif isinstance(idx, list):
        idx = torch.tensor(idx, dtype=torch.int64)
    elif isinstance(idx, slice):
        idx = torch.arange(idx.stop)[idx]
    elif isinstance(idx, np.ndarray):
        idx = torch.from_numpy(idx)
    if idx.dtype == torch.bool:
        idx = torch.where(idx)[0]
    assert idx.dtype is torch.int64, f"Expected LongTensor but got {idx.dtype} instead."
    return idx


class CSRData(object):

    def __init__(self, pointers, *args, dense=False, is_index_value=None):
        self.pointers = CSRData._sorted_indices_to_pointers(pointers) if dense else pointers
        self.values = [*args] if len(args) > 0 else None
        if is_index_value is None or len(is_index_value) == 0:
            self.is_index_value = torch.zeros(self.num_values, dtype=torch.bool)
        else:
            self.is_index_value = torch.as_tensor(is_index_value, dtype=torch.bool)

    # -- bookkeeping --------------------------------------------------------
    @property
    def device(self):
        return self.pointers.device

    @property
    def num_groups(self):
        return self.pointers.shape[0] - 1

    @property
    def num_values(self):
        return len(self.values) if self.values is not None else 0

    @property
    def num_items(self):
        return self.pointers[-1].item()

    def __len__(self):
        return self.num_groups

    @staticmethod
    def get_batch_type():
        return CSRBatch

    def clone(self):
        """Shallow copy: new container, same tensors."""
        out = copy.copy(self)
        out.pointers = self.pointers
        out.values = None if self.values is None else [
            v.clone() if isinstance(v, CSRData) else v for v in self.values]
        return out

    def _moved(self, device, non_blocking):
        out = self.clone()
        out.pointers = out.pointers.to(device, non_blocking=non_blocking)
        for i in range(out.num_values):
            v = out.values[i]
            out.values[i] = v._moved(device, non_blocking) if isinstance(v, CSRData) \
                else v.to(device, non_blocking=non_blocking)
        return out

    def to(self, device):
        """Copy on ``device``.  Device -> host copies of all arrays are queued
        together into pinned memory and waited for once (a blocking copy per
        array into fresh pageable memory costs several times as much)."""
        device = torch.device(device)
        to_host = device.type == 'cpu' and self.pointers.is_cuda
        out = self._moved(device, to_host)
        if to_host:
            torch.cuda.current_stream(self.pointers.device).synchronize()
        return out

    def cpu(self):
        return self.to('cpu')

    def cuda(self):
        return self.to('cuda')

    def debug(self):
        assert self.pointers[0] == 0, "The first pointer element must always be 0."
        assert torch.all(self.pointers[1:] - self.pointers[:-1] >= 0), \
            "pointer indices must be increasing."
        if self.values is not None:
            assert isinstance(self.values, list), "Values must be held in a list."
            assert all(len(v) == self.num_items for v in self.values), \
                "All value objects must have the same size."
            for v in self.values:
                if isinstance(v, CSRData):
                    v.debug()
            assert self.is_index_value.dtype == torch.bool
            assert self.is_index_value.shape[0] == self.num_values

    # -- dense indices <-> pointers ----------------------------------------
    @staticmethod
    def _is_sorted(a):
        return torch.all(a[:-1] <= a[1:])

    @staticmethod
    def _sorted_indices_to_pointers(indices):
        """Sorted dense group indices -> pointers (csr.py:158-172)."""
        assert indices.dim() == 1, "Only 1D indices are accepted."
        assert indices.shape[0] >= 1, "At least one group index is required."
        assert CSRData._is_sorted(indices), "Indices must be sorted in increasing order."
        dev = indices.device
        jumps = torch.where(indices[1:] > indices[:-1])[0] + 1
        return torch.cat([
            torch.zeros(1, dtype=torch.int64, device=dev), jumps,
            torch.tensor([indices.shape[0]], dtype=torch.int64, device=dev)])

    def insert_empty_groups(self, group_indices, num_groups=None):
        """In place: existing group i becomes group ``group_indices[i]`` (sorted)
        of a larger CSR whose other groups are empty (csr.py:197-229)."""
        assert self.num_groups == group_indices.shape[0], \
            "New group indices must correspond to the existing number of groups"
        assert CSRData._is_sorted(group_indices), "New group indices must be sorted."
        top = int(group_indices.max()) + 1
        num_groups = top if num_groups is None else max(top, int(num_groups))
        g = group_indices.to(self.device)
        lo = torch.cat([g.new_tensor([-1]), g])
        hi = torch.cat([g, g.new_tensor([num_groups])])
        self.pointers = self.pointers.repeat_interleave(hi - lo)
        return self

    def reindex_groups(self, group_indices, num_groups=None):
        """csr.py:174-195."""
        order = torch.argsort(group_indices)
        return self[order].insert_empty_groups(group_indices[order], num_groups=num_groups)

    @staticmethod
    def _index_select_pointers(pointers, indices):
        """Pointers of the selected groups + gather indices for the values
        (csr.py:235-264)."""
        assert indices.max() <= pointers.shape[0] - 2
        dev = pointers.device
        sizes = pointers[indices + 1] - pointers[indices]
        new_ptr = torch.cat([
            torch.zeros(1, dtype=pointers.dtype, device=dev), torch.cumsum(sizes, 0)])
        total = int(new_ptr[-1])
        val_idx = torch.arange(total, device=dev)
        val_idx = val_idx - new_ptr[:-1].repeat_interleave(sizes) \
            + pointers[indices].repeat_interleave(sizes)
        return new_ptr, val_idx

    def __getitem__(self, idx):
        idx = tensor_idx(idx).to(self.device)
        out = self.clone()
        if idx.shape[0] == 0:
            out.pointers = torch.zeros(1, dtype=torch.int64, device=self.device)
            out.values = [v[[]] for v in self.values]
        else:
            out.pointers, val_idx = CSRData._index_select_pointers(self.pointers, idx)
            out.values = [v[val_idx] for v in self.values]
        return out

    def __repr__(self):
        info = [f"{k}={getattr(self, k)}" for k in ['num_groups', 'num_items', 'device']]
        return f"{self.__class__.__name__}({', '.join(info)})"


class CSRBatch(CSRData):
    """Concatenation of CSRData items that can be split again
    (csr.py:305-479)."""
    __csr_type__ = CSRData

    def __init__(self, pointers, *args, dense=False, is_index_value=None):
        super(CSRBatch, self).__init__(
            pointers, *args, dense=dense, is_index_value=is_index_value)
        self.__sizes__ = None

    @property
    def batch_pointers(self):
        if self.__sizes__ is None:
            return None
        return torch.cumsum(torch.cat((torch.zeros(1, dtype=torch.int64), self.__sizes__.cpu())), dim=0)

    @property
    def batch_items_sizes(self):
        return self.__sizes__

    @property
    def num_batch_items(self):
        return len(self.__sizes__) if self.__sizes__ is not None else 0

    def _moved(self, device, non_blocking):
        out = super(CSRBatch, self)._moved(device, non_blocking)
        out.__sizes__ = self.__sizes__.to(device) if self.__sizes__ is not None else None
        return out

    @staticmethod
    def from_csr_list(csr_list):
        assert isinstance(csr_list, list) and len(csr_list) > 0
        first = csr_list[0]
        assert isinstance(first, CSRData), "All provided items must be CSRData objects."
        csr_type = type(first)
        assert all(isinstance(c, csr_type) for c in csr_list), \
            "All provided items must have the same class."
        dev = first.device
        assert all(c.device == dev for c in csr_list), \
            "All provided items must be on the same device."
        assert all(c.num_values == first.num_values for c in csr_list), \
            "All provided items must have the same number of values."
        is_index_value = first.is_index_value
        assert all(torch.equal(c.is_index_value, is_index_value) for c in csr_list), \
            "All provided items must have the same is_index_value."

        item_offsets = np.cumsum([0] + [c.num_items for c in csr_list[:-1]])
        pointers = torch.cat(
            [torch.zeros(1, dtype=torch.int64, device=dev)]
            + [c.pointers[1:] + int(o) for c, o in zip(csr_list, item_offsets)])

        values = []
        for i in range(first.num_values):
            parts = [c.values[i] for c in csr_list]
            if isinstance(parts[0], CSRData):
                values.append(CSRBatch.from_csr_list(parts))
            elif is_index_value[i]:
                # index-valued entries are shifted by the running max + 1
                shift = np.cumsum(
                    [0] + [int(v.max()) + 1 if v.shape[0] > 0 else 0 for v in parts[:-1]])
                values.append(torch.cat([v + int(s) for v, s in zip(parts, shift)], dim=0))
            else:
                values.append(torch.cat(parts, dim=0))

        batch = csr_type.get_batch_type()(
            pointers, *values, dense=False, is_index_value=is_index_value)
        batch.__sizes__ = torch.tensor([c.num_groups for c in csr_list], dtype=torch.int64)
        batch.__csr_type__ = csr_type
        return batch

    def to_csr_list(self):
        if self.__sizes__ is None:
            raise RuntimeError(
                'Cannot reconstruct CSRData data list from batch because the '
                'batch object was not created using `CSRBatch.from_csr_list()`.')
        gp = self.batch_pointers.to(self.device)
        ip = self.pointers[gp]
        n = self.num_batch_items
        pointers = [self.pointers[gp[i]:gp[i + 1] + 1] - ip[i] for i in range(n)]
        per_value = []
        for i in range(self.num_values):
            bv = self.values[i]
            if isinstance(bv, CSRData):
                per_value.append(bv.to_csr_list())
            elif self.is_index_value[i]:
                per_value.append([
                    bv[ip[j]:ip[j + 1]] - (bv[:ip[j]].max() + 1 if j > 0 else 0)
                    for j in range(n)])
            else:
                per_value.append([bv[ip[j]:ip[j + 1]] for j in range(n)])
        per_item = [list(v) for v in zip(*per_value)]
        return [self.__csr_type__(p, *v, dense=False, is_index_value=self.is_index_value)
                for p, v in zip(pointers, per_item)]

    def __getitem__(self, idx):
        sub = super(CSRBatch, self).__getitem__(idx)
        return self.__csr_type__(
            sub.pointers, *sub.values, dense=False, is_index_value=sub.is_index_value)

    def __repr__(self):
        info = [f"{k}={getattr(self, k)}"
                for k in ['num_batch_items', 'num_groups', 'num_items', 'device']]
        return f"{self.__class__.__name__}({', '.join(info)})"
