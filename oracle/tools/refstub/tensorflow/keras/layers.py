class _L:
    def __init__(self, *a, **k):
        pass


class Dense(_L):
    pass


class Flatten(_L):
    pass


class Conv2D(_L):
    pass


class Dropout(_L):
    pass


class Softmax(_L):
    pass
