"""Output helpers of the command lines (mirror of pydca/dca_utilities/dca_utilities.py:
create_directories :9-27, get_dca_output_file_path :29-57, *_param_metadata :109-169,
mfdca_residue_repr_metadata :172-201, write_sorted_dca_scores :236-266, write_couplings_csv
:293-325, write_fields_csv :328-359, write_single_site_freqs :362-395, write_pair_site_freqs
:398-436).  File layout and number formatting match the reference."""
import errno
import logging
import os

from ..fasta_reader import fasta_reader

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


def mfdca_residue_repr_metadata(biomolecule):
    """dca_utilities.py:172-201: integer -> letter table, five pairs per line."""
    metadata_list = ['# RESIDUES IDENTIFICATION']
    pairs = sorted(fasta_reader.res_to_char(biomolecule).items(), key=lambda k: k[0])
    for i in range(int(len(pairs) / 5) + 1):
        row = pairs[i * 5:(i + 1) * 5]
        row.insert(0, '# ')
        metadata_list.append(''.join(map(str, row)))
    return metadata_list


def write_couplings_csv(file_name, couplings, metadata=None):
    """dca_utilities.py:293-325: 1-based site pair, then the (q-1)^2 shifted couplings."""
    logger.info('\n\tSaving couplings to file:\n\t{}'.format(file_name))
    with open(file_name, 'w') as fh:
        fh.write('#' + '=' * 70 + '\n')
        if metadata:
            for data in metadata:
                fh.write('{}\n'.format(data))
            fh.write('#' + '=' * 70 + '\n')
        for site_pair, couplings_ij in couplings:
            fh.write('{},{}'.format(site_pair[0] + 1, site_pair[1] + 1))
            for c in couplings_ij:
                fh.write(',{}'.format(c))
            fh.write('\n')
    return None


def write_fields_csv(file_name, fields, metadata=None):
    """dca_utilities.py:328-359.  As in the reference, the field rows are only written when
    metadata is given (its loop sits inside the metadata branch); every caller passes metadata."""
    logger.info('\n\tSaving fields to file:\n\t{}'.format(file_name))
    with open(file_name, 'w') as fh:
        fh.write('#{}\n'.format(70 * '='))
        if metadata is not None:
            for data in metadata:
                fh.write('{}\n'.format(data))
            fh.write('#{}\n'.format(70 * '='))
            for site, site_fields in fields:
                fh.write('{}'.format(site + 1))
                for fia in site_fields:
                    fh.write(',{}'.format(fia))
                fh.write('\n')
    return None


def write_single_site_freqs(file_name, fi, seqs_len=None, num_site_states=None, metadata=None):
    """dca_utilities.py:362-395."""
    logger.info('\n\tSaving single site frequencies to file:\n\t{}'.format(file_name))
    with open(file_name, 'w') as fh:
        fh.write('#' + '=' * 70 + '\n')
        if metadata:
            for data in metadata:
                fh.write('{}\n'.format(data))
            fh.write('# Below, the First integer refers to the site, the \n'
                     '# Second the residue at that site, and the Third is the \n'
                     '# frequency. Residue numbers are mapped as shown above.\n')
            fh.write('#' + '=' * 70 + '\n')
        for i in range(seqs_len):
            for a in range(num_site_states):
                fh.write('{},{},{}\n'.format(i + 1, a + 1, fi[i, a]))
    return None


def write_pair_site_freqs(file_name, fij, seqs_len=None, num_site_states=None, metadata=None):
    """dca_utilities.py:398-436 (gap state excluded)."""
    logger.info('\n\tSaving pair site frequencies (gaps are excluded) to file: \n\t{}'.format(file_name))
    with open(file_name, 'w') as fh:
        fh.write('#' + '=' * 70 + '\n')
        if metadata:
            for data in metadata:
                fh.write('{}\n'.format(data))
            fh.write('# Below, the First and Second integers refer to sites, the \n'
                     '# Third and Fourth residues, and the Last one is frequency for pairs.\n'
                     '# Residue numbers are mapped as shown above.\n')
            fh.write('#' + '=' * 70 + '\n')
        pair_counter = 0
        for i in range(seqs_len - 1):
            for j in range(i + 1, seqs_len):
                for a in range(num_site_states - 1):
                    for b in range(num_site_states - 1):
                        fh.write('{},{},{},{},{}\n'.format(i + 1, j + 1, a + 1, b + 1, fij[pair_counter, a, b]))
                pair_counter += 1
    return None


def write_trimmed_msa(file_name, msa_trimmer=None, columns_to_remove=None, metadata=None):
    """dca_utilities.py:581-607: FASTA, one line per sequence, the listed columns dropped."""
    logger.info('\n\tWritting trimmed MSA in to file {}'.format(file_name))
    drop = set(columns_to_remove)
    with open(file_name, 'w') as fh:
        for record in msa_trimmer.alignment_data:
            trimmed_seq = [record.seq[i] for i in range(len(record.seq)) if i not in drop]
            fh.write('>{}\n{}\n'.format(record.id, ''.join(trimmed_seq)))
    return None
