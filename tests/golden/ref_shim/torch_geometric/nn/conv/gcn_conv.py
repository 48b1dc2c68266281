def gcn_norm(*a, **k):
    raise NotImplementedError("stand-in: not on the SignNet/BasisNet path")
