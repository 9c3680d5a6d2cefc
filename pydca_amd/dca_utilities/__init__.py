from .dca_utilities import (create_directories, get_dca_output_file_path, mfdca_param_metadata,  # noqa: F401
                            plmdca_param_metadata, write_sorted_dca_scores)
