"""Scalars that stay on the device until somebody needs their value.

The reference trainer reads ``loss.item()`` and the six loss terms after EVERY step, only to add them to running sums that are divided and
printed once per epoch (train/vector_field_nerf_train.py:262-280).  With plain floats that is a device synchronisation per step: the
host cannot issue step t + 1 before step t has finished, and every step pays the host's lead-in (0.25 ms at the reference's 1 024-ray
batches, 12 % of the step).  ``DeferredScalar`` is what ``loss.item()`` and ``losses_dict[key]`` return on the HIP path instead: a
number whose value is entry ``i`` of a small device vector.  Adding two of them adds the VECTORS on the device (one tiny launch per pair
of vectors, whatever the number of entries read) and gives another deferred scalar; EVERY other use — division, comparison, ``float()``,
formatting, ``repr``, arithmetic with a Python number — waits for the device, reads the value and continues as the ``float`` it is.
So the reference loop's accumulation runs without a synchronisation and its ``average_losses[key] /= len(dataloader)`` at the end of the
epoch turns every entry into a plain float.  ``loss.DEFERRED_SCALARS = False`` switches it off (floats at once, as in rounds 1-4)."""
from __future__ import annotations

import numbers
from typing import List, Optional

import torch


class DeviceScalars:
    """A small device vector of scalars, read back at most once."""

    __slots__ = ("dev", "_host", "_last")

    def __init__(self, dev: torch.Tensor) -> None:
        self.dev = dev
        self._host: Optional[List[float]] = None
        self._last = None                 # (other, self + other): the entries of one step are added pairwise to the same running vector

    def values(self) -> List[float]:
        if self._host is None:
            self._host = self.dev.detach().double().tolist()      # (synchronises)
            self.dev = None
        return self._host

    def plus(self, other: "DeviceScalars") -> "DeviceScalars":
        last = self._last
        if last is not None and last[0] is other:
            return last[1]
        if self._host is not None or other._host is not None or self.dev.device != other.dev.device or self.dev.shape != other.dev.shape:
            a, b = self.values(), other.values()
            out = DeviceScalars.__new__(DeviceScalars)
            out.dev, out._host, out._last = None, [x + y for x, y in zip(a, b)] if len(a) == len(b) else None, None
            if out._host is None:
                raise ValueError("deferred scalars of different vectors cannot be added entry by entry")
        else:
            with torch.no_grad():
                out = DeviceScalars(self.dev.double() + other.dev.double())      # running sums in float64, as the trainer's Python floats are
        self._last = (other, out)
        return out


def _binary(name):
    def op(self, other):
        return getattr(float(self), name)(float(other) if isinstance(other, DeferredScalar) else other)
    op.__name__ = name
    return op


class DeferredScalar:
    """Entry ``i`` of a ``DeviceScalars`` vector, behaving as the float it will be."""

    __slots__ = ("_h", "_i")

    def __init__(self, holder: DeviceScalars, index: int) -> None:
        self._h, self._i = holder, int(index)

    # -- the one deferred operation -----------------------------------------------------------------
    def __add__(self, other):
        if isinstance(other, DeferredScalar) and other._i == self._i:
            return DeferredScalar(self._h.plus(other._h), self._i)
        return float(self) + (float(other) if isinstance(other, DeferredScalar) else other)

    def __radd__(self, other):
        return other + float(self)

    __iadd__ = __add__

    # -- everything else is the float ---------------------------------------------------------------
    def __float__(self) -> float:
        return float(self._h.values()[self._i])

    def item(self) -> float:
        return float(self)

    def __int__(self) -> int:
        return int(float(self))

    def __bool__(self) -> bool:
        return bool(float(self))

    def __repr__(self) -> str:
        return repr(float(self))

    __str__ = __repr__

    def __format__(self, spec: str) -> str:
        return format(float(self), spec)

    def __hash__(self) -> int:
        return hash(float(self))

    def __neg__(self):
        return -float(self)

    def __pos__(self):
        return float(self)

    def __abs__(self):
        return abs(float(self))

    def __round__(self, n=None):
        return round(float(self), n)

    def __trunc__(self):
        return float(self).__trunc__()

    def __floor__(self):
        return float(self).__floor__()

    def __ceil__(self):
        return float(self).__ceil__()

    def is_integer(self) -> bool:
        return float(self).is_integer()

    @property
    def real(self):
        return float(self)

    @property
    def imag(self):
        return 0.0

    def conjugate(self):
        return float(self)

    # -- leaving the process / entering numpy: as the float ------------------------------------------
    def __reduce__(self):
        """pickle / copy.deepcopy / multiprocessing queues: the value travels as a plain ``float`` (the device vector does not)."""
        return (float, (float(self),))

    def __copy__(self):
        return float(self)

    def __deepcopy__(self, memo):
        return float(self)

    def __array__(self, dtype=None, copy=None):
        """``np.asarray(x)``, ``np.asarray([x, y])``, ``np.mean([...])``: a float64 scalar array, so that a list of them becomes a float64
        array and not an object array."""
        import numpy as np
        return np.asarray(float(self), dtype=dtype or np.float64)

    def as_integer_ratio(self):
        return float(self).as_integer_ratio()

    def hex(self) -> str:
        return float(self).hex()


for _n in ("__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__", "__rtruediv__", "__floordiv__", "__rfloordiv__", "__mod__", "__rmod__",
           "__pow__", "__rpow__", "__divmod__", "__rdivmod__", "__eq__", "__ne__", "__lt__", "__le__", "__gt__", "__ge__"):
    setattr(DeferredScalar, _n, _binary(_n))
numbers.Real.register(DeferredScalar)


def resolve(obj):
    """``obj`` with every deferred scalar inside it (dicts, lists, tuples, nested) replaced by its ``float`` — what a logger does before it
    hands a step's numbers to code that insists on the exact type (``json.dumps`` without ``default=``, ``isinstance(x, float)``,
    msgpack).  One device read-back per distinct step vector, however many entries are read."""
    if isinstance(obj, DeferredScalar):
        return float(obj)
    if isinstance(obj, dict):
        return {k: resolve(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(resolve(v) for v in obj)
    return obj


def json_default(obj):
    """``json.dumps(payload, default=deferred.json_default)`` (``default=float`` works as well): the encoder's hook for types it does not
    know; deferred scalars leave as floats, anything else is refused as json itself would."""
    if isinstance(obj, DeferredScalar):
        return float(obj)
    raise TypeError(f"Object of type {type(obj).__name__} is not JSON serializable")


class LossTensor(torch.Tensor):
    """The loss of a step: a tensor like any other (``backward()``, arithmetic, ``float()``), whose ``item()`` hands out the deferred
    scalar of the step's term vector when there is one."""

    __torch_function__ = torch._C._disabled_torch_function_impl       # results of operations on it are plain tensors

    def item(self):
        s = getattr(self, "_vfn_scalar", None)
        if s is not None:
            return s
        return torch.Tensor.item(self)


def as_loss(total: torch.Tensor, holder: Optional[DeviceScalars], index: int) -> torch.Tensor:
    """``total`` (graph attached or not) with ``item()`` deferred to entry ``index`` of ``holder``."""
    if holder is None:
        return total
    out = total.as_subclass(LossTensor)
    out._vfn_scalar = DeferredScalar(holder, index)
    return out
