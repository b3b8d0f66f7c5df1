class Discrete(object):
    def __init__(self, n):
        self.n = int(n)
