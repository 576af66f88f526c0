"""pip install -e .  — builds csrc/libuavx.so with hipcc (gfx950) in-tree, like `make -C .../csrc` does.
(The reference's README installs with `pip install -e .` too; its setup.py pins gym / pygame / tensorflow,
none of which this package needs.)"""
import os
import subprocess

from setuptools import find_packages, setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "gym_uav_collision_avoidance_amd", "csrc")


class BuildWithHip(build_py):
    def run(self):
        subprocess.check_call(["make", "-C", CSRC])
        super().run()


setup(
    name="gym_uav_collision_avoidance_amd",
    version="0.1.0",
    description="MI355X-native batched implementation of the gym_uav_collision_avoidance step/reset path",
    packages=find_packages(include=["gym_uav_collision_avoidance_amd", "gym_uav_collision_avoidance_amd.*"]),
    # compat/: the opt-in alias package `gym_uav_collision_avoidance` (install_alias()); shipped as data, never a top-level package
    package_data={"gym_uav_collision_avoidance_amd": ["csrc/*.so", "csrc/*.hip", "csrc/*.hpp", "csrc/Makefile",
                                                      "compat/gym_uav_collision_avoidance/*.py",
                                                      "compat/gym_uav_collision_avoidance/envs/*.py"]},
    install_requires=["numpy", "torch"],
    cmdclass={"build_py": BuildWithHip},
)
