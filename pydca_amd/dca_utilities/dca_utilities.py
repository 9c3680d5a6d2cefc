"""Output helpers touched by `compute_fn` (mirror of pydca/dca_utilities/dca_utilities.py:
create_directories :9-27, get_dca_output_file_path :29-57, *_param_metadata :109-169,
write_sorted_dca_scores :236-266).  File layout and number formatting match the reference."""
import errno
import logging
import os

logger = logging.getLogger(__name__)


def create_directories(the_path):
    try:
        os.makedirs(the_path)
    except OSError as e:
        if e.errno != errno.EEXIST:
            logger.error('Unable to create directory using path {}'.format(the_path))
            raise
    return None


def get_dca_output_file_path(output_dir, msa_file_name, prefix='', postfix=''):
    msa_file_root, _ext = os.path.splitext(os.path.basename(msa_file_name))
    return os.path.join(output_dir, prefix.strip() + msa_file_root.strip() + postfix.strip())


def mfdca_param_metadata(mfdca_instance):
    return [
        '# PARAMETERS USED FOR THIS COMPUTATION: ',
        '#      Sequence type: {}'.format(mfdca_instance.biomolecule),
        '#      Total number of sequences in alignment data: {}'.format(mfdca_instance.num_sequences),
        '#      Length of sequences in alignment data: {}'.format(mfdca_instance.sequences_len),
        '#      Effective number of sequences: {}'.format(mfdca_instance.effective_num_sequences),
        '#      Value of sequence identity: {}'.format(mfdca_instance.sequence_identity),
        '#      Value of relative pseudocount: {}'.format(mfdca_instance.pseudocount),
    ]


def plmdca_param_metadata(plmdca_instance):
    return [
        '# PARAMETERS USED FOR THIS COMPUTATION: ',
        '#\tSequence type: {}'.format(plmdca_instance.biomolecule),
        '#\tTotal number of sequences in alignment data: {}'.format(plmdca_instance.num_sequences),
        '#\tLength of sequences in alignment data: {}'.format(plmdca_instance.sequences_len),
        '#\tValue of sequence identity: {}'.format(plmdca_instance.sequence_identity),
        '#\tlambda_h: {}'.format(plmdca_instance.lambda_h),
        '#\tlambda_J: {}'.format(plmdca_instance.lambda_J),
        '#\tNumber of gradient decent iterations: {}'.format(plmdca_instance.max_iterations),
    ]


def write_sorted_dca_scores(file_name, sorted_DI, metadata=None, score_type=None):
    logger.info('\n\tWriting DCA scores to file {}'.format(file_name))
    with open(file_name, 'w') as fh:
        fh.write('#' + '=' * 70 + '\n')
        if metadata:
            for line in metadata:
                fh.write('{}\n'.format(line))
        fh.write('# The First and Second columns represent sites and the'
                 '\n# Third column is {} DCA score\n'.format(score_type))
        fh.write('#' + '=' * 70 + '\n')
        for pair, score in sorted_DI:
            i, j = pair
            fh.write('{0:<7} {1:<14} {2:<35}\n'.format(i + 1, j + 1, score))
    return None
