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


def copy_u(u, out):
    return ("edge", lambda g: g.ndata[u][g.src], out)


def copy_edge(e, out):          # older name of copy_e (layers/transformer.py:97)
    return copy_e(e, out)


def src_mul_edge(src, edge, out):    # older name of u_mul_e (layers/transformer.py:96)
    return u_mul_e(src, edge, out)
