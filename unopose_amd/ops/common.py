"""Helpers shared by the operator families of `unopose_amd.ops`: differentiable mode, fall-back notes, cache keys, the mutation epoch."""
import itertools

import torch

from . import _state as st


def _c(x):
    return x if x.is_contiguous() else x.contiguous()


class differentiable:
    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        self.prev, st._DIFF = st._DIFF, bool(self.on)
        return self

    def __exit__(self, *a):
        st._DIFF = self.prev
        return False


def is_differentiable():
    return st._DIFF


_fallbacks_seen = set()


def note_fallback(site, why):
    """A shape / configuration the hand-written kernels do not take runs the torch composite instead -- still on the GPU, never
    silently: one RuntimeWarning per (site, reason) names it."""
    key = (site, why)
    if key not in _fallbacks_seen:
        _fallbacks_seen.add(key)
        import warnings

        warnings.warn(f"unopose_amd.ops.{site}: {why} -> torch composite (not a hand-written kernel)", RuntimeWarning, stacklevel=3)


def _params_key(mod, *extra):
    """Cache key of state derived from a module's tensors: version and address of EVERY parameter and buffer of the module, so that an
    in-place edit of any of them -- a bias alone included -- rebuilds the derived state.  The (owner dict, name)
    slots are listed once per module (its structure does not change; a REPLACED tensor is seen through the slot) -- walking
    `mod.parameters()` on every call cost ~7 us x 147 calls per forward (scripts/host_profile.py)."""
    slots = mod.__dict__.get("_pk_slots")
    if slots is None:
        slots = [(m._parameters, k) for m in mod.modules() for k in m._parameters] + [(m._buffers, k) for m in mod.modules() for k in m._buffers]
        mod.__dict__["_pk_slots"] = slots
    key = []
    for d, k in slots:
        t = d[k]
        if t is not None:
            key.append((t._version, t.data_ptr()))
    return tuple(key) + extra


_SPLIT_MEMO = []  # [(key, source tensor, split tensor)], newest first


_MUTATION_EPOCH = [0]  # bumped by every wrapper that writes a tensor through its raw pointer (in place, `out=`): see split_f32


def note_mutation():
    """A kernel is about to write an existing tensor through `ptr()` (torch's version counter does not see that): results remembered
    for tensors of an earlier epoch are dropped."""
    _MUTATION_EPOCH[0] += 1
    _SPLIT_MEMO.clear()


def clear_split_memo():
    """end of a forward: the remembered operands (and the source tensors they pin) are released"""
    _SPLIT_MEMO.clear()


def _aligned16(x):
    """16-byte aligned storage once contiguous (the streaming kernels move float4 / float2 per lane)."""
    return (x.data_ptr() % 16 == 0) if x.is_contiguous() else True  # (_c() copies a non-contiguous view into a fresh, aligned buffer)


def _no_autograd():
    """The hand-written fp32 fast paths return tensors WITHOUT a grad_fn: they are taken only where autograd is not recording
    (no_grad / inference mode -- every eval entry point of this package).  With gradients enabled the call falls through to the
    torch composite (or to `linear_train` in differentiable mode), so eval-mode gradient use (pose refinement, saliency) stays
    correct instead of silently losing its graph."""
    return not torch.is_grad_enabled()


def _f32_path(x):
    return x.is_cuda and x.dtype == torch.float32 and not st._DIFF and not torch.is_autocast_enabled() and _no_autograd()


def _own_f32(x):
    return x.is_cuda and not st._DIFF and st.USE_F32X3 and _no_autograd()


def _own_glue(x):
    """the small fp32 steps between the kernels on own kernels (csrc/glue.hip, round 5): eval on the GPU, nothing recorded by autograd"""
    return x.is_cuda and not st._DIFF and _no_autograd()
