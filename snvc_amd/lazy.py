"""A cost volume that is built when somebody looks at it.

The reference's call sequence is ``volume = build_cost_volume(left, right, shift, 1); cost = model(volume)``.  The left
half of that CONCAT volume repeats the left feature on every disparity plane, and the first 3D convolution is the only
consumer: ``GlobalStack.forward`` does not need the 1.48 GB tensor at all (models/stereo_volume.py::forward_pair builds
only the warped right half and folds the left half into three depth-class planes -- DESIGN 4.1b).  ``LazyCostVolume`` lets
the reference's own operator API reach that path: ``build_cost_volume`` returns it under ``torch.no_grad()``; it has the
volume's shape / dtype / device, ``GlobalStack.forward`` takes it apart again, and ANY other use -- an aten operator, a
kernel of this library asking for ``data_ptr()``, printing -- builds the real volume first (once) and carries on with
that: the observable values are those of the eager volume, bit for bit.
"""
import torch
from torch.utils._pytree import tree_map


import threading


class _Consumer(threading.local):
    """Weakref to the GlobalStack that consumed the last lazy volume on the fused path IN THIS THREAD (set by GlobalStack.forward): the
    next build_cost_volume(...) of the thread starts that model's first-layer prep in front of its own host sync
    (GlobalStack.lazy_prefetch).  Thread-local (r6): the reference's op is stateless and is called from DataParallel worker threads, one
    per device (tools/inference_agnostic.py:472) -- a worker must never start another worker's model."""
    ref = None

    # the r5 spelling CONSUMER[0] keeps working
    def __getitem__(self, i):
        return self.ref

    def __setitem__(self, i, value):
        self.ref = value


CONSUMER = _Consumer()


class LazyCostVolume(torch.Tensor):
    @staticmethod
    def __new__(cls, left, right, shift, downsample, build, spacing="unknown", prefetch=None):
        n, c, h, w = left.shape
        shape = (n, 2 * c, shift.shape[1], h // downsample, w // downsample)
        r = torch.Tensor._make_wrapper_subclass(cls, shape, dtype=left.dtype, device=left.device, requires_grad=False)
        r._sources = (left, right, shift, downsample)
        r._build = build
        r._real = None
        r._real_version = None
        r._spacing = spacing
        r._source_versions = (left._version, right._version, shift._version)
        r._own_version = r._version      # in-place aten operators on the wrapper bump ITS counter (above __torch_dispatch__)
        r._prefetch = prefetch           # (weakref to the model, its paused step, the weight version it was started with) or None
        return r

    def take_prefetch(self, model):
        """The paused step ``build_cost_volume`` started for ``model`` (GlobalStack.lazy_prefetch), once: None if there is none, it
        belongs to another model or stream, or the model's first-layer weights changed since.  (Whether the prep buffers the paused
        step queued are still ITS OWN is the step's business: it re-checks the model's prep epoch when resumed and starts over if another
        call used them in between -- GlobalStack._forward_pair_steps.)"""
        pre, self._prefetch = self._prefetch, None
        if pre is None:
            return None
        ref, gen, wver = pre
        w = model.conv1[0][0].weight
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if ref() is not model or wver != (w.data_ptr(), w._version, stream):
            gen.close()
            return None
        return gen

    @property
    def spacing(self):
        """What build_cost_volume's one look at the shift array found: (q, m0) for uniformly spaced planes (m0 + d) / q,
        None for any other array, "unknown" if it did not classify it."""
        return self._spacing

    # ---- what GlobalStack.forward uses
    @property
    def is_materialized(self):
        return self._real is not None

    @property
    def sources(self):
        """(left, right, shift, downsample) as given to build_cost_volume."""
        return self._sources

    @property
    def sources_unchanged(self):
        """No in-place write to left / right / shift since build_cost_volume was called."""
        left, right, shift, _ = self._sources
        return (left._version, right._version, shift._version) == self._source_versions

    @property
    def is_pristine(self):
        """The volume still equals build_cost_volume(*sources): never built, or built and only LOOKED at since (its version
        counter has not moved; nobody wrote through an aten operator) -- and the sources themselves are untouched.  Then
        GlobalStack.forward may take the fused path from the sources: same values as the eager volume would give."""
        if not self.sources_unchanged:
            return False
        if self._version != self._own_version:
            return False
        return self._real is None or self._real._version == self._real_version

    def materialize(self) -> torch.Tensor:
        if self._prefetch is not None:       # somebody looks at the volume itself: the step started for its consumer is dropped
            self._prefetch[1].close()
            self._prefetch = None
        if self._real is None:
            if not self.sources_unchanged:
                raise RuntimeError("LazyCostVolume: left / right / shift were modified in place after build_cost_volume(...) and "
                                   "before the volume was first used; the volume the call described can no longer be built "
                                   "(clone the inputs, or call build_cost_volume with autograd enabled for the eager tensor)")
            left, right, shift, ds = self._sources
            self._real = self._build(left, right, shift, ds)
            self._real_version = self._real._version
        return self._real

    # ---- everything else sees the real tensor
    def data_ptr(self):
        return self.materialize().data_ptr()

    def __repr__(self):
        return f"LazyCostVolume(shape={tuple(self.shape)}, materialized={self.is_materialized})"

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        def unwrap(t):
            return t.materialize() if isinstance(t, LazyCostVolume) else t
        return func(*tree_map(unwrap, args), **tree_map(unwrap, kwargs or {}))
