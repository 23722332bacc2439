"""TEST DOUBLE of ``stable_baselines3.common.vec_env`` (see the package docstring): the abstract VecEnv contract of
SB3 1.6 -- constructor (num_envs, observation_space, action_space), abstract reset / step_async / step_wait / close /
get_attr / set_attr / env_method / env_is_wrapped / seed, concrete step / unwrapped / _get_indices."""
from abc import ABC, abstractmethod


class VecEnv(ABC):
    metadata = {"render.modes": ["human", "rgb_array"]}

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space
        self.sb3_vecenv_init_called = True     # lets the test see that the subclass ran this constructor

    @abstractmethod
    def reset(self): ...

    @abstractmethod
    def step_async(self, actions): ...

    @abstractmethod
    def step_wait(self): ...

    @abstractmethod
    def close(self): ...

    @abstractmethod
    def get_attr(self, attr_name, indices=None): ...

    @abstractmethod
    def set_attr(self, attr_name, value, indices=None): ...

    @abstractmethod
    def env_method(self, method_name, *method_args, indices=None, **method_kwargs): ...

    @abstractmethod
    def env_is_wrapped(self, wrapper_class, indices=None): ...

    @abstractmethod
    def seed(self, seed=None): ...

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    @property
    def unwrapped(self):
        return self

    def _get_indices(self, indices):
        if indices is None:
            indices = range(self.num_envs)
        elif isinstance(indices, int):
            indices = [indices]
        return indices


def is_vecenv_wrapped(venv, vec_wrapper_class):
    return False
