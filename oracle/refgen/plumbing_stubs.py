"""Inert stand-ins for the *plumbing* modules the reference imports but this image lacks.

TEST INFRASTRUCTURE ONLY -- used by oracle/refgen/gen_golden.py in the build container to import
the read-only reference at /root/reference and record golden vectors.  Nothing here carries
arithmetic: `overrides` is a decorator, `loguru`/`wandb` are logging, and `torch_nlp_utils`
(git dependency pinned at 2afbbce7..., poetry.lock:1276-1295, not vendored) supplies a class
registry, a params dict and data-pipeline base classes.  All maths executed by the generator is the
reference's own code running on stock torch.

Registry semantics mirrored: `X.register(name)` stores the subclass in a
registry keyed by the class `register` was called on; `X.from_params(type=..., **p)` pops `type`,
looks the subclass up and calls the subclass's own `from_params` override when it has one.
"""
import sys
import types


def _module(name):
    mod = types.ModuleType(name)
    sys.modules[name] = mod
    return mod


class _Quiet:
    """Object whose every attribute is a callable returning itself (loguru.logger stand-in)."""

    def __getattr__(self, _name):
        return self

    def __call__(self, *args, **kwargs):
        return self


class FromParams:
    @classmethod
    def from_params(cls, **params):
        return cls(**params)


class Registrable(FromParams):
    _registries = {}

    @classmethod
    def register(cls, name):
        table = Registrable._registries.setdefault(cls, {})

        def deco(subclass):
            table[name] = subclass
            return subclass

        return deco

    @classmethod
    def by_name(cls, name):
        return Registrable._registries[cls][name]

    @classmethod
    def from_params(cls, **params):
        table = Registrable._registries.get(cls)
        if table is not None and "type" in params:
            params = dict(params)
            sub = table[params.pop("type")]
            own = sub.__dict__.get("from_params")
            if own is not None:
                return sub.from_params(**params)
            # walk the MRO for a from_params override below Registrable
            for klass in sub.__mro__:
                if klass in (Registrable, FromParams):
                    break
                if "from_params" in klass.__dict__:
                    return klass.__dict__["from_params"].__get__(None, sub)(**params)
            return sub(**params)
        return cls(**params)


class Params(dict):
    pass


class Vocabulary:
    def __init__(self, size=100):
        self._size = size

    def token_to_index(self, token, namespace=None):
        return {"<eos>": 2, "<sos>": 1, "@@UNKNOWN@@": 3}.get(token, 3)

    def get_vocab_size(self, namespace=None):
        return self._size


class _Any(Registrable):
    def __init__(self, *args, **kwargs):
        pass


def install():
    ov = _module("overrides")
    ov.overrides = lambda f: f

    lg = _module("loguru")
    lg.logger = _Quiet()

    wb = _module("wandb")
    wb.watch = lambda *a, **k: None
    wb.Table = _Quiet()
    wb.init = lambda *a, **k: None
    wb.log = lambda *a, **k: None

    root = _module("torch_nlp_utils")
    common = _module("torch_nlp_utils.common")
    common.FromParams, common.Registrable, common.Params = FromParams, Registrable, Params
    data = _module("torch_nlp_utils.data")
    data.Vocabulary = Vocabulary
    for name in ("DatasetReader", "CollateBatch", "DataIterator", "Batch", "Namespace"):
        setattr(data, name, type(name, (_Any,), {}))
    metrics = _module("torch_nlp_utils.metrics")
    metrics.Metric = type("Metric", (), {})
    callbacks = _module("torch_nlp_utils.callbacks")
    callbacks.EarlyStopping = type("EarlyStopping", (_Any,), {})
    callbacks.SaveCheckpoint = type("SaveCheckpoint", (_Any,), {})
    root.common, root.data, root.metrics, root.callbacks = common, data, metrics, callbacks
