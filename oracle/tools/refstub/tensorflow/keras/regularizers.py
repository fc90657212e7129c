"""regularisers only add training-loss terms: nothing at inference"""


def l2(*a, **k):
    return None
