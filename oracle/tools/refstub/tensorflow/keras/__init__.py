"""tensorflow.keras stand-in: Model with the two weight-file formats the reference loads."""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

from . import layers, regularizers  # noqa: F401

_H5DUMP = "/opt/conda/bin/h5dump"


def _h5_strings(path, attr):
    out = subprocess.run([_H5DUMP, "-a", attr, path], check=True, capture_output=True, text=True).stdout
    data = out[out.index("DATA {"):]
    return re.findall(r'"([^"]*)"', data)


def _h5_dataset(path, dset):
    with tempfile.NamedTemporaryFile(suffix=".bin") as tmp:
        subprocess.run([_H5DUMP, "-d", dset, "-b", "LE", "-o", tmp.name, path], check=True, stdout=subprocess.DEVNULL)
        return np.fromfile(tmp.name, dtype="<f4")


class _Status:
    def expect_partial(self):
        return self

    def assert_consumed(self):
        return self


class Model:
    """Subclassed-model behaviour the reference relies on: layers are tracked in attribute-assignment order,
    `__call__` casts floating inputs to float32 (Keras autocast) and runs `call()` outside training."""

    def __init__(self, *a, **k):
        object.__setattr__(self, "_layers", [])
        object.__setattr__(self, "_pending", None)

    def __setattr__(self, name, value):
        if isinstance(value, layers.Layer):
            value._attr = name
            self._layers.append(value)
        object.__setattr__(self, name, value)

    @property
    def layers(self):
        return list(self._layers)

    # ----------------------------------------------------------------------------------------- forward
    def __call__(self, inputs, training=None):
        import tensorflow as tf

        def cast(a):
            a = np.asarray(a)
            # Keras autocast: floating inputs become float32 (f16 -> f32 exact, f64 -> f32 rounds); the layers here
            # then compute in tf.COMPUTE_DTYPE
            return a.astype(np.float32).astype(tf.COMPUTE_DTYPE) if a.dtype.kind == "f" else a
        x = [cast(a) for a in inputs] if isinstance(inputs, (list, tuple)) else cast(inputs)
        if self._pending is not None:
            self._build_by_running(x)
            self._apply(self._pending)
            object.__setattr__(self, "_pending", None)
        out = self.call(x)
        if tf.RETURN_F32:
            conv = lambda o: tf._t(np.asarray(o).astype(np.float32))
            return tuple(conv(o) for o in out) if isinstance(out, tuple) else conv(out)
        return out

    def _build_by_running(self, x):
        self.call(x)                                  # every layer creates zero weights of the right shape

    def build(self, input_shape=None):
        shape = [d if d is not None else 1 for d in input_shape]
        self._build_by_running(np.zeros(shape, np.float32))

    # ----------------------------------------------------------------------------------------- weights
    def load_weights(self, path):
        if path.endswith(".h5"):
            self._load_h5(path)
        else:
            # TF-format restore is deferred until the variables exist (first call), as in TensorFlow
            object.__setattr__(self, "_pending", self._read_tf(path))
            if all(l.built for l in self._layers if l.has_weights):
                self._apply(self._pending)
                object.__setattr__(self, "_pending", None)
        return _Status()

    @staticmethod
    def _read_tf(prefix):
        here = os.path.dirname(os.path.abspath(__file__))
        tools = os.path.abspath(os.path.join(here, "..", "..", ".."))
        if tools not in sys.path:
            sys.path.insert(0, tools)
        from convert_weights import read_tf_checkpoint          # the TF-bundle parser
        return read_tf_checkpoint(prefix)

    def _apply(self, tensors):
        """object-graph matching of a TF checkpoint: `<attribute>/<variable>/.ATTRIBUTES/VARIABLE_VALUE`"""
        n = 0
        for lay in self._layers:
            if not lay.has_weights:
                continue
            for var in ("kernel", "bias"):
                key = "%s/%s/.ATTRIBUTES/VARIABLE_VALUE" % (lay._attr, var)
                if key not in tensors:
                    raise KeyError("checkpoint has no %s" % key)
                cur = getattr(lay, var)
                if tuple(tensors[key].shape) != tuple(cur.shape):
                    raise ValueError("%s: checkpoint %s vs model %s" % (key, tensors[key].shape, cur.shape))
                setattr(lay, var, tensors[key].astype(np.float32))
                n += 1
        assert n > 0

    def _load_h5(self, path):
        """Keras `load_weights_from_hdf5_group`: the file's `layer_names` that own weights are matched BY ORDER with the
        model's layers that own weights; inside a layer by the order of `weight_names` (kernel, bias)."""
        mine = [l for l in self._layers if l.has_weights]
        if not all(l.built for l in mine):
            raise ValueError("load_weights(.h5) on a model that has not been built (call it once first)")
        names = []
        for ln in _h5_strings(path, "layer_names"):
            wn = _h5_strings(path, "/%s/weight_names" % ln)
            if wn:
                names.append((ln, wn))
        if len(names) != len(mine):
            raise ValueError("h5 has %d weighted layers, the model %d" % (len(names), len(mine)))
        for lay, (ln, wn) in zip(mine, names):
            vals = [_h5_dataset(path, "/%s/%s" % (ln, w)) for w in wn]
            if len(vals) != 2:
                raise ValueError("layer %s: %d weights" % (ln, len(vals)))
            for var, v in zip(("kernel", "bias"), vals):
                cur = getattr(lay, var)
                if v.size != cur.size:
                    raise ValueError("%s/%s: %d values vs %s" % (ln, var, v.size, cur.shape))
                setattr(lay, var, v.reshape(cur.shape).astype(np.float32))
