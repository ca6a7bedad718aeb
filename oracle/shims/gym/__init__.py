"""ORACLE — TEST INFRASTRUCTURE ONLY.

Minimal stand-in for the `gym` package (not installed offline) so the UNMODIFIED reference at
/root/reference imports: it only uses `gym.Env` as a base class, `gym.make(id)` and
`gym.envs.registration.register(id=, entry_point=)` (crowd_sim/__init__.py:1-6,
crowd_nav/train.py:76, crowd_nav/test.py:64).
"""
import importlib

from . import envs  # noqa: F401
from .envs.registration import register, registry


class Env(object):
    metadata = {}

    def reset(self, *a, **k):
        raise NotImplementedError

    def step(self, *a, **k):
        raise NotImplementedError

    def render(self, *a, **k):
        raise NotImplementedError


def make(env_id):
    entry = registry[env_id]
    if callable(entry):
        return entry()
    mod_name, cls_name = entry.split(':')
    return getattr(importlib.import_module(mod_name), cls_name)()
