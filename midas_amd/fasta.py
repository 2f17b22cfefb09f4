"""Minimal FASTA reader standing in for Bio.SeqIO.parse(infile, 'fasta') at midas/run/snps.py:59-62 and :78-79:
`rec.id` is the header up to the first whitespace, `str(rec.seq)` the concatenated sequence lines."""


def parse(handle):
    """Yield (id, seq) per record."""
    rec_id, chunks = None, []
    for line in handle:
        if line.startswith('>'):
            if rec_id is not None:
                yield rec_id, ''.join(chunks)
            header = line[1:].strip()
            rec_id = header.split()[0] if header else ''
            chunks = []
        elif rec_id is not None:
            chunks.append(''.join(line.split()))
    if rec_id is not None:
        yield rec_id, ''.join(chunks)
