class Summary:
    def __init__(self, *a, **k):
        pass
