"""ForwardKL agent on MI355X (mirrors agents/ForwardKL.py:13-97 + agents/network/forwardkl_network.py).

Identical to ReverseKL around the network (the reference's two manager files differ only in the network class); the
policy update is the forward-KL action integral of forwardkl_network.py:159-190.
"""
from .base_agent import BaseAgent
from .ReverseKL import KL_Network_Manager


class ForwardKL_Network_Manager(KL_Network_Manager):
    KIND = "forward"


class ForwardKL(BaseAgent):
    def __init__(self, config):
        network_manager = ForwardKL_Network_Manager(config)
        super(ForwardKL, self).__init__(config, network_manager)
