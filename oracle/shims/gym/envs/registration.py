registry = {}


def register(id, entry_point, **kwargs):  # noqa: A002 - keeps gym's keyword name
    registry[id] = entry_point
