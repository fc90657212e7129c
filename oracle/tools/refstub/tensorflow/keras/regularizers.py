def l2(*a, **k):
    return None
