from .modeling import InstanceSam, Sam
from .build_sam import build_instance_sam, build_sam
