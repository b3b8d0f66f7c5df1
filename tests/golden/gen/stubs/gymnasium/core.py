class Env(object):
    metadata = {}
    action_space = None
    observation_space = None

    @property
    def unwrapped(self):
        return self

    def step(self, action):
        raise NotImplementedError

    def reset(self, *, seed=None, options=None):
        raise NotImplementedError


class Wrapper(Env):
    def __init__(self, env):
        self.env = env

    @property
    def unwrapped(self):
        return self.env.unwrapped
