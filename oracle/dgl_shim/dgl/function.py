"""TEST INFRASTRUCTURE ONLY -- dgl.function builtins the reference names
(model_zoo.py:41 copy_src/sum, model_zoo.py:95 src_mul_edge/sum)."""


class _Msg:
    def __init__(self, kind, src=None, edge=None, out=None):
        self.kind, self.src, self.edge, self.out = kind, src, edge, out


class _Red:
    def __init__(self, kind, msg, out):
        self.kind, self.msg, self.out = kind, msg, out


def copy_src(src, out):
    return _Msg("copy_src", src=src, out=out)


def src_mul_edge(src, edge, out):
    return _Msg("src_mul_edge", src=src, edge=edge, out=out)


def sum(msg, out):  # noqa: A001 - mirrors dgl.function.sum
    return _Red("sum", msg, out)
