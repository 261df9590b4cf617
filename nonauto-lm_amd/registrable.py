"""Minimal class registry mirroring the slice of torch_nlp_utils.common the reference uses
(`Registrable.register / by_name / from_params`, `FromParams`).  The reference's dependency
(Nemexur/torch-nlp-utils @ 2afbbce7, poetry.lock:1276-1295) is plumbing only; this is a
from-scratch equivalent so configs written for the reference (`{"type": "nice", ...}`) build the
MI355X modules."""
from typing import Dict, Type


class FromParams:
    @classmethod
    def from_params(cls, **params):
        return cls(**params)


class Registrable(FromParams):
    _tables: Dict[type, Dict[str, type]] = {}

    @classmethod
    def register(cls, name: str):
        table = Registrable._tables.setdefault(cls, {})

        def decorator(subclass: Type):
            if name in table and table[name] is not subclass:
                raise ValueError(f"{name!r} is already registered for {cls.__name__}")
            table[name] = subclass
            return subclass

        return decorator

    @classmethod
    def by_name(cls, name: str):
        try:
            return Registrable._tables[cls][name]
        except KeyError:
            known = sorted(Registrable._tables.get(cls, {}))
            raise KeyError(f"{name!r} is not a registered {cls.__name__}; known: {known}") from None

    @classmethod
    def list_available(cls):
        return sorted(Registrable._tables.get(cls, {}))

    @classmethod
    def from_params(cls, **params):
        if cls in Registrable._tables and "type" in params:
            params = dict(params)
            sub = cls.by_name(params.pop("type"))
            for klass in sub.__mro__:
                if klass in (Registrable, FromParams):
                    break
                if "from_params" in klass.__dict__:
                    return klass.__dict__["from_params"].__get__(None, sub)(**params)
            return sub(**params)
        return cls(**params)
