from .matrices import AVAILABLE_ROTATIONS, AVAILABLE_UNITS, translation_matrix, rotation_matrix, shear_matrix, \
    scale_matrix, transform_matrix
from .general import compute_prefilter_workgroup_dims, compute_elementwise_launch_dims, get_available_devices, \
    switch_to_device, parse_device, compute_post_transform_dimensions
