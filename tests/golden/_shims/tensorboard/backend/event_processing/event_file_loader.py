class EventFileLoader:
    def __init__(self, *a, **k):
        raise NotImplementedError
