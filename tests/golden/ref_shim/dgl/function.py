"""Stand-in for dgl.function: the four built-ins the reference's GatedGCN layer uses (layers/gatedgcn_layer.py:51-56),
restated from DGL's documented semantics: message functions produce an edge field, `sum` reduces it over each node's in-edges."""


def u_add_v(lhs, rhs, out):
    return ("edge", lambda g: g.ndata[lhs][g.src] + g.ndata[rhs][g.dst], out)


def u_mul_e(lhs, rhs, out):
    return ("edge", lambda g: g.ndata[lhs][g.src] * g.edata[rhs], out)


def copy_e(e, out):
    return ("edge", lambda g: g.edata[e], out)


def sum(msg, out):  # noqa: A001
    return ("reduce_sum", msg, out)
