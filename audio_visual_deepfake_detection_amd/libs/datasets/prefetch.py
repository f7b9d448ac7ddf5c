"""Host -> HBM staging for the dataset loaders.

The reference moves a batch to the GPU synchronously inside `preprocessing` (`batched_inputs.to(self.device)`,
av_fd_no_recon.py:477: 9.4 MB of pageable fp32 per clip).  Here the loader's items carry the raw per-modality
`streams` ([T_s, C_s] numpy arrays: 8.6 MB per 768-frame clip before K1 expands them on the device); this wrapper keeps
`depth` batches in flight: every array is copied into a recycled pinned buffer and sent with an asynchronous H2D copy on
a dedicated HIP stream, so the copy of batch i+1 overlaps the compute of batch i.  The consumer's stream waits on the
batch's event only.  `model.preprocessing` accepts the resulting device tensors unchanged."""
import collections

import numpy as np
import torch


class _PinnedPool:
    def __init__(self):
        self.free = collections.defaultdict(list)

    def get(self, n):
        cap = 1 << max(12, (int(n) - 1).bit_length())          # power-of-two buckets, >= 4 KiB of floats
        lst = self.free[cap]
        return lst.pop() if lst else torch.empty(cap, dtype=torch.float32).pin_memory()

    def put(self, buf):
        self.free[buf.numel()].append(buf)


class DevicePrefetcher:
    """for batch in DevicePrefetcher(loader, device): losses = model(batch) ..."""

    def __init__(self, loader, device, depth=2, keys=("streams",)):
        self.loader, self.device, self.depth, self.keys = loader, torch.device(device), max(1, depth), keys
        assert self.device.type == "cuda", "staging targets a HIP device"
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.pool = _PinnedPool()

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        held = []
        with torch.cuda.stream(self.copy_stream):
            out = []
            for item in batch:
                item = dict(item)
                for k in self.keys:
                    if k not in item:
                        continue
                    dev_list = []
                    for a in item[k]:
                        a = np.ascontiguousarray(a, dtype=np.float32)
                        pin = self.pool.get(a.size)
                        view = pin[:a.size].view(a.shape)
                        view.copy_(torch.from_numpy(a))
                        dev_list.append(view.to(self.device, non_blocking=True))
                        held.append(pin)
                    item[k] = dev_list
                out.append(item)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return out, ev, held

    def __iter__(self):
        it = iter(self.loader)
        inflight = collections.deque()
        done = False
        while True:
            while not done and len(inflight) < self.depth:
                try:
                    inflight.append(self._stage(next(it)))
                except StopIteration:
                    done = True
            if not inflight:
                return
            batch, ev, held = inflight.popleft()
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for item in batch:
                for k in self.keys:
                    for t in item.get(k, ()):
                        t.record_stream(cur)
            ev.synchronize()                 # the pinned buffers of this batch may be reused from here on
            for pin in held:
                self.pool.put(pin)
            yield batch
