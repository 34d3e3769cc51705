from .agent import RobotAgent  # noqa: F401  (reference module path: social_gym/src/robot_agent.py)
