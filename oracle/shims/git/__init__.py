"""ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for gitpython (crowd_nav/train.py:9,57)."""


class _Obj(object):
    hexsha = '0' * 40


class _Head(object):
    object = _Obj()


class Repo(object):
    def __init__(self, *a, **k):
        self.head = _Head()
