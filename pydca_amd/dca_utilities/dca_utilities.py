"""File output of the command lines.  The TEXT these functions write is the contract with the reference
(pydca/dca_utilities/dca_utilities.py: metadata lines :109-201, score files :236-266, couplings / fields CSV :293-359,
frequency files :362-436, trimmed alignments :581-607) and is reproduced character for character -- rule lines of
'#' + 70 '=', the metadata wording, 1-based sites, '{0:<7} {1:<14} {2:<35}' score rows, comma-separated rows without
padding.  How the text is produced is this module's own: every writer builds its header and hands an iterator of
rows to one routine that streams them out."""
import errno
import logging
import os

from ..fasta_reader import fasta_reader

logger = logging.getLogger(__name__)

_RULE = '#' + '=' * 70


def create_directories(the_path):
    """mkdir -p; an existing directory is fine (dca_utilities.py:9-27)."""
    try:
        os.makedirs(the_path)
    except OSError as exc:
        if exc.errno != errno.EEXIST:
            logger.error('cannot create the output directory {}'.format(the_path))
            raise


def get_dca_output_file_path(output_dir, msa_file_name, prefix='', postfix=''):
    """<output_dir>/<prefix><alignment file name without extension><postfix> (dca_utilities.py:29-57)."""
    stem = os.path.splitext(os.path.basename(msa_file_name))[0]
    return os.path.join(output_dir, ''.join(part.strip() for part in (prefix, stem, postfix)))


def _stream(file_name, header, rows, what):
    """Writes the header lines and then the rows (already formatted, without line ends) to file_name."""
    logger.info('\n\twriting {} to {}'.format(what, file_name))
    with open(file_name, 'w') as fh:
        fh.writelines(line + '\n' for line in header)
        fh.writelines(row + '\n' for row in rows)


def _described(label_width_prefix, instance, fields):
    return ['# PARAMETERS USED FOR THIS COMPUTATION: '] + [
        '#{}{}: {}'.format(label_width_prefix, label, getattr(instance, attr)) for label, attr in fields]


def mfdca_param_metadata(mfdca_instance):
    return _described('      ', mfdca_instance, (
        ('Sequence type', 'biomolecule'),
        ('Total number of sequences in alignment data', 'num_sequences'),
        ('Length of sequences in alignment data', 'sequences_len'),
        ('Effective number of sequences', 'effective_num_sequences'),
        ('Value of sequence identity', 'sequence_identity'),
        ('Value of relative pseudocount', 'pseudocount')))


def plmdca_param_metadata(plmdca_instance):
    return _described('\t', plmdca_instance, (
        ('Sequence type', 'biomolecule'),
        ('Total number of sequences in alignment data', 'num_sequences'),
        ('Length of sequences in alignment data', 'sequences_len'),
        ('Value of sequence identity', 'sequence_identity'),
        ('lambda_h', 'lambda_h'),
        ('lambda_J', 'lambda_J'),
        ('Number of gradient decent iterations', 'max_iterations')))      # sic: the reference's wording, part of the file format


def mfdca_residue_repr_metadata(biomolecule):
    """'# RESIDUES IDENTIFICATION' and the integer -> letter table as Python tuples, five per line, plus the line the
    reference's range(len // 5 + 1) adds at the end (empty but for '# ' when the table divides by five)."""
    table = sorted(fasta_reader.res_to_char(biomolecule).items())
    lines = ['# RESIDUES IDENTIFICATION']
    for first in range(0, 5 * (len(table) // 5 + 1), 5):
        lines.append('# ' + ''.join(map(str, table[first:first + 5])))
    return lines


def write_sorted_dca_scores(file_name, sorted_DI, metadata=None, score_type=None):
    header = [_RULE] + list(metadata or []) + [
        '# The First and Second columns represent sites and the',
        '# Third column is {} DCA score'.format(score_type), _RULE]
    rows = ('{0:<7} {1:<14} {2:<35}'.format(pair[0] + 1, pair[1] + 1, score) for pair, score in sorted_DI)
    _stream(file_name, header, rows, 'ranked scores')


def _csv(prefix_values, values):
    # '{}'.format(v), not str(v): numpy scalars of the two print differently (a float32 is widened by format)
    return ','.join('{}'.format(v) for v in list(prefix_values) + list(values))


def write_couplings_csv(file_name, couplings, metadata=None):
    """One row per site pair: i, j (1-based) and the (q-1)^2 couplings of the pair."""
    header = [_RULE] + (list(metadata) + [_RULE] if metadata else [])
    rows = (_csv((pair[0] + 1, pair[1] + 1), block) for pair, block in couplings)
    _stream(file_name, header, rows, 'couplings')


def write_fields_csv(file_name, fields, metadata=None):
    """One row per site: i (1-based) and its q-1 fields.  Without metadata only the rule line is written: in the
    reference the rows are produced inside the metadata branch (dca_utilities.py:346-357) and every caller passes it."""
    header = [_RULE]
    rows = ()
    if metadata is not None:
        header += list(metadata) + [_RULE]
        rows = (_csv((site + 1,), site_fields) for site, site_fields in fields)
    _stream(file_name, header, rows, 'fields')


def write_single_site_freqs(file_name, fi, seqs_len=None, num_site_states=None, metadata=None):
    header = [_RULE]
    if metadata:
        header += list(metadata) + [
            '# Below, the First integer refers to the site, the ',
            '# Second the residue at that site, and the Third is the ',
            '# frequency. Residue numbers are mapped as shown above.', _RULE]
    rows = ('{},{},{}'.format(i + 1, a + 1, fi[i, a]) for i in range(seqs_len) for a in range(num_site_states))
    _stream(file_name, header, rows, 'single-site frequencies')


def write_pair_site_freqs(file_name, fij, seqs_len=None, num_site_states=None, metadata=None):
    """Pair frequencies without the gap state; fij is indexed [pair in (0,1),(0,2),... order, a, b]."""
    header = [_RULE]
    if metadata:
        header += list(metadata) + [
            '# Below, the First and Second integers refer to sites, the ',
            '# Third and Fourth residues, and the Last one is frequency for pairs.',
            '# Residue numbers are mapped as shown above.', _RULE]
    states = range(num_site_states - 1)
    pairs = ((i, j) for i in range(seqs_len - 1) for j in range(i + 1, seqs_len))
    rows = ('{},{},{},{},{}'.format(i + 1, j + 1, a + 1, b + 1, fij[p, a, b])
            for p, (i, j) in enumerate(pairs) for a in states for b in states)
    _stream(file_name, header, rows, 'pair-site frequencies (gap state left out)')


def write_trimmed_msa(file_name, msa_trimmer=None, columns_to_remove=None, metadata=None):
    """FASTA with one line per sequence; the listed columns are left out of every record."""
    dropped = frozenset(columns_to_remove)
    rows = ('>{}\n{}'.format(record.id, ''.join(ch for col, ch in enumerate(record.seq) if col not in dropped))
            for record in msa_trimmer.alignment_data)
    _stream(file_name, [], rows, 'the trimmed alignment')
