"""DeviceFeatureStore -- the feature hand-off into training without the [L, h(h+2)] host tensor.

The reference materialises the subgraph features of EVERY link of a split on the host (datasets/elph.py:207-208: 356 M x 15
floats = 21 GB for ogbl-citation2) and each training / inference batch then does

    subgraph_features = data.subgraph_features[sf_indices].to(device)        # runners/train.py:58-60, inference.py:119-120

i.e. a host gather + a PCIe copy per batch.  With the sketch tables resident in HBM the same rows are cheaper to RECOMPUTE per
batch than to gather on the host and ship (tools/probe_feature_store.py, 2.66 M links, two boxes: 19-23 / 531-700 / 1 045-1 177 M
pairs/s at batches of 1 024 / 65 536 / 1 M against 1.5-14 / 26-370 / 21-30 M pairs/s for the host-tensor hand-off: 1.6-47x),
and nothing of size L x F ever exists.  `DeviceFeatureStore` is the object to put where `HashDataset.subgraph_features` is:
it answers the indexing the reference's loops and dataset code perform on that tensor

    store[idx_tensor] / store[list] / store[i] / store[a:b]     -> float32 [n, F] ON THE DEVICE, computed by ss_pair_features
    store[rows, cols]                                            -> the same, then the column selection
    store.shape, len(store), store.dtype, store.device, store.to(device)        (`.to` of the result is then a no-op)
    store[store < 0] = 0          (datasets/elph.py:214-215, floor_sf)           -> sets the kernel's floor flag
    store[:, [4, 5]] = 0          (datasets/elph.py:218-222, use_zero_one=False) -> remembered, applied to every batch
    torch.save(store, path)       (datasets/elph.py:212-213)                     -> saves the materialised CPU tensor

Rows are bit-identical to `ElphHashes.get_subgraph_features(links, ...)[idx]`: a pair's features depend on nothing but its own
sketch rows (one 16-lane group per pair, fixed reduction order), so the composition of a batch cannot change them
(tests/test_gpu_parity.py::test_device_feature_store).  There is no CPU path: the store needs the HIP engine.
"""
import torch

from .hashing import ElphHashes, _compute_device


class _NegativeEntries(object):
    """what `store < 0` returns: a token the floor_sf statements of datasets/elph.py:214-217 hand back to the store"""

    def __init__(self, store):
        self.store = store


class DeviceFeatureStore(object):
    def __init__(self, elph_hashes, links, hash_table, cards, degrees=None, device=None, batch_size=11000000):
        """elph_hashes: the ElphHashes that built (hash_table, cards); links: int [L, 2] (any device: a device copy is kept,
        16 B per link); degrees: optional float [N] -> rows carry BUDDY's degree-normalised copy too ([n, 2F])"""
        if not isinstance(elph_hashes, ElphHashes):
            raise TypeError('DeviceFeatureStore needs the ElphHashes engine that built the tables')
        if links.dim() != 2 or links.size(1) != 2:
            raise ValueError('links must have shape [L, 2]')
        first = hash_table.get(1) if hasattr(hash_table, 'get') else None
        self.device = torch.device(device) if device is not None else _compute_device(getattr(first, 'mh_u32', None), cards, links)
        self._eh, self._table, self._cards, self._degrees = elph_hashes, hash_table, cards, degrees
        self._links = links.to(device=self.device, dtype=torch.int64).contiguous()
        self._batch = int(batch_size)
        self._floor = bool(elph_hashes.floor_sf)
        self._zero_cols = []
        nf = elph_hashes.max_hops * (elph_hashes.max_hops + 2)
        self._width = nf * (2 if degrees is not None else 1)
        self.dtype = torch.float32

    # ---- tensor-like surface ---------------------------------------------------------------------------------------
    @property
    def shape(self):
        return torch.Size((self._links.size(0), self._width))

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 2

    def __len__(self):
        return self._links.size(0)

    def to(self, *args, **kwargs):
        """stays lazy on its device (the reference only ever calls `.to(device)` on the batch it has just indexed)"""
        target = kwargs.get('device', args[0] if args and not isinstance(args[0], torch.dtype) else None)
        if target is not None and torch.device(target).type != self.device.type:
            return self.materialise().to(*args, **kwargs)
        return self

    def cpu(self):
        return self.materialise().cpu()

    def __lt__(self, other):
        if other != 0:
            raise NotImplementedError('only `store < 0` (the floor_sf statement of datasets/elph.py:214-215) is lazy; materialise() first')
        return _NegativeEntries(self)

    def __setitem__(self, key, value):
        if isinstance(key, _NegativeEntries) and key.store is self and float(value) == 0.0:
            self._floor = True                      # store[store < 0] = 0
            return
        if (isinstance(key, tuple) and len(key) == 2 and key[0] == slice(None) and float(value) == 0.0
                and isinstance(key[1], (list, tuple, int))):
            cols = [key[1]] if isinstance(key[1], int) else list(key[1])
            self._zero_cols = sorted(set(self._zero_cols) | {int(c) % self._width for c in cols})   # store[:, [4, 5]] = 0
            return
        raise NotImplementedError('DeviceFeatureStore only records the two in-place edits HashDataset performs '
                                  '(floor and column knock-out, datasets/elph.py:214-222); materialise() for anything else')

    def __getitem__(self, key):
        if isinstance(key, _NegativeEntries):       # after the floor there are none (datasets/elph.py:216-217 sums them)
            if self._floor:
                return torch.zeros(0, dtype=self.dtype, device=self.device)
            full = self.materialise()
            return full[full < 0]
        cols = None
        if isinstance(key, tuple):
            if len(key) != 2:
                raise IndexError('too many indices for a 2-D feature store')
            key, cols = key
        rows = self._rows(key)
        out = self._compute(rows)
        if isinstance(key, int):
            out = out[0]
        return out if cols is None else out[..., cols]

    def __reduce_ex__(self, proto):
        """torch.save(store, path) writes what the reference would have written: the materialised CPU tensor (and
        torch.load's default weights_only unpickler reads it back as an ordinary tensor)"""
        return self.materialise().cpu().__reduce_ex__(proto)

    # ---- computation -------------------------------------------------------------------------------------------------
    def _rows(self, key):
        L = self._links.size(0)
        if isinstance(key, int):
            if not -L <= key < L:
                raise IndexError(f'index {key} is out of bounds for {L} links')
            return self._links[key:key + 1] if key >= 0 else self._links[L + key:L + key + 1]
        if isinstance(key, slice):
            return self._links[key]
        idx = torch.as_tensor(key)
        if idx.dtype == torch.bool:
            raise NotImplementedError('boolean masks are not lazy; materialise() first')
        return self._links[idx.to(device=self.device, dtype=torch.int64)]

    def _compute(self, links):
        eh = self._eh
        chunks = []
        for s in range(0, max(links.size(0), 1), self._batch):
            part = links[s:s + self._batch]
            feats, _ = eh._pair_kernel(part, self._table, self._cards, degrees=self._degrees, floor_sf=self._floor)
            chunks.append(feats)
        out = chunks[0] if len(chunks) == 1 else torch.cat(chunks, dim=0)
        if self._zero_cols:
            out[:, self._zero_cols] = 0
        return out

    def materialise(self, out_device=None):
        """the full [L, F] tensor (what the reference holds), on the store's device unless told otherwise"""
        out = self._compute(self._links)
        return out if out_device is None else out.to(out_device)
