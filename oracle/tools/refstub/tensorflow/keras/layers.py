"""Keras layers as plain numpy (container-only; see tensorflow/__init__.py).  Inference semantics only."""
import numpy as np

SELU_SCALE = 1.0507009873554804934193349852946
SELU_ALPHA = 1.6732632423543772848170429916717


def _dt():
    import tensorflow as tf
    return tf.COMPUTE_DTYPE


def _wrap(a):
    import tensorflow as tf
    return tf._t(a)


def _softmax(x):
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def _activation(name):
    if name is None or name == "linear":
        return lambda x: x
    if name == "selu":
        return lambda x: SELU_SCALE * np.where(x > 0, x, SELU_ALPHA * np.expm1(np.minimum(x, 0)))
    if name == "sigmoid":
        return lambda x: 1.0 / (1.0 + np.exp(-x))
    if name == "softmax":
        return _softmax
    raise NotImplementedError("activation %r" % (name,))


class Layer:
    has_weights = False

    def __init__(self, *a, name=None, **k):
        self.name = name
        self.built = False
        self._attr = None


class Dense(Layer):
    has_weights = True

    def __init__(self, units, activation=None, use_bias=True, name=None, **k):
        super().__init__(name=name)
        assert use_bias
        self.units = int(units)
        self.act = _activation(activation)

    def __call__(self, x):
        x = np.asarray(x)
        if not self.built:
            self.kernel = np.zeros((x.shape[-1], self.units), np.float32)
            self.bias = np.zeros(self.units, np.float32)
            self.built = True
        dt = _dt()
        y = x.astype(dt) @ self.kernel.astype(dt) + self.bias.astype(dt)
        return _wrap(self.act(y))


class Conv2D(Layer):
    """NHWC input, HWIO kernel, cross-correlation; TF padding arithmetic: 'valid' out = floor((in - k)/s) + 1;
    'same' out = ceil(in/s), pad_total = max((out-1)*s + k - in, 0), pad_before = pad_total // 2"""
    has_weights = True

    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", activation=None, use_bias=True,
                 name=None, **k):
        super().__init__(name=name)
        assert use_bias
        self.filters = int(filters)
        self.ks = (int(kernel_size[0]), int(kernel_size[1]))
        self.st = (int(strides[0]), int(strides[1]))
        self.padding = padding.lower()
        assert self.padding in ("same", "valid")
        self.act = _activation(activation)

    def __call__(self, x):
        x = np.asarray(x)
        n, h, w, c = x.shape
        kh, kw = self.ks
        sh, sw = self.st
        if not self.built:
            self.kernel = np.zeros((kh, kw, c, self.filters), np.float32)
            self.bias = np.zeros(self.filters, np.float32)
            self.built = True
        dt = _dt()
        x = x.astype(dt)
        if self.padding == "same":
            oh, ow = -(-h // sh), -(-w // sw)
            ph = max((oh - 1) * sh + kh - h, 0)
            pw = max((ow - 1) * sw + kw - w, 0)
            x = np.pad(x, ((0, 0), (ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)))
        else:
            oh, ow = (h - kh) // sh + 1, (w - kw) // sw + 1
        ker = self.kernel.astype(dt)
        y = np.zeros((n, oh, ow, self.filters), dt)
        for i in range(kh):
            for j in range(kw):
                patch = x[:, i:i + (oh - 1) * sh + 1:sh, j:j + (ow - 1) * sw + 1:sw, :]
                y += patch @ ker[i, j]
        y += self.bias.astype(dt)
        return _wrap(self.act(y))


class Flatten(Layer):
    def __call__(self, x):
        x = np.asarray(x)
        return _wrap(x.reshape(x.shape[0], -1))


class Dropout(Layer):
    def __init__(self, rate, **k):
        super().__init__()
        self.rate = rate

    def __call__(self, x, training=None):
        assert not training
        return x


class Softmax(Layer):
    def __init__(self, axis=-1, **k):
        super().__init__()
        assert axis == -1

    def __call__(self, x):
        return _wrap(_softmax(np.asarray(x)))
