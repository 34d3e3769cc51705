from .agent import HumanAgent  # noqa: F401  (reference module path: social_gym/src/human_agent.py)
