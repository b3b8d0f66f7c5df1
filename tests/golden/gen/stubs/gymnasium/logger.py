INFO = 20
DISABLED = 50


def set_level(level):
    pass


def info(msg, *args):
    pass
