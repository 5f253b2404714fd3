"""`mmmm.models` surface of the reference: MMMMForCausalLM / build, Sam / build_sam, InstanceSam / build_instance_sam."""
from .mmmm import MMMMForCausalLM, build
from .segvol import InstanceSam, Sam, build_instance_sam, build_sam
