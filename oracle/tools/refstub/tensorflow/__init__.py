"""numpy `tensorflow` stand-in (container-only test tooling; never shipped, never imported by the product).

It lets the reference's own model classes (/root/reference/nanocaller_src/model_architect*.py) and worker
loops (snpCaller.caller, indelCaller.indel_run) be IMPORTED AND EXECUTED UNCHANGED in the build container,
where TensorFlow is absent: the layer classes in tensorflow.keras.layers are plain-numpy implementations of
the documented Keras semantics (NHWC Conv2D as cross-correlation with HWIO kernels, TF 'same'/'valid'
padding arithmetic, Dense y = xW + b, Flatten in C order, SELU constants, softmax over the last axis,
Dropout = identity outside training), and Model.load_weights reads the reference's REAL checkpoint files
(TF bundle: by attribute path; Keras H5: by layer order, as Keras does).  What runs is therefore the
reference's own `call()` wiring and its own weight files; what is restated here is only the arithmetic of
each Keras layer.  Arithmetic is float64 on float32 inputs/weights (COMPUTE_DTYPE), so goldens made with it
are within float32 rounding of what TensorFlow's float32 kernels return.
"""
import numpy as np

from . import keras  # noqa: F401

COMPUTE_DTYPE = np.float64     # internal arithmetic of every layer
RETURN_F32 = False             # True: Model.__call__ rounds its outputs to float32 (what TF hands back)


class Tensor(np.ndarray):
    """ndarray with the `.numpy()` accessor eager tensors have (snpCaller.py:183)"""

    def numpy(self):
        return np.asarray(self)


def _t(a):
    return np.asarray(a).view(Tensor)


def concat(values, axis, name=None):
    vals = [np.asarray(v) for v in values]
    if len({v.dtype for v in vals}) != 1:
        raise TypeError("tf.concat: mixed dtypes %s (Keras casts Model inputs to float32 before call())"
                        % [v.dtype for v in vals])
    return _t(np.concatenate(vals, axis=axis))
