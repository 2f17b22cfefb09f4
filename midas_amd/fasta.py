"""Minimal FASTA reader standing in for Bio.SeqIO.parse(infile, 'fasta') at midas/run/snps.py:59-62 and :78-79:
`rec.id` is the header up to the first whitespace, `str(rec.seq)` the concatenated sequence lines."""


def parse_bytes(data: bytes):
    """The same records from the file's bytes, ids as str and sequences as bytes: no text decoding of hundreds of megabases
    that go to the device as bytes anyway."""
    if data.startswith(b'>'):
        data = b'\n' + data
    for part in data.split(b'\n>')[1:]:
        header, _, body = part.partition(b'\n')
        header = header.strip()
        yield (header.split()[0].decode('latin-1') if header else ''), b''.join(body.split())


def parse(handle):
    """Yield (id, seq) per record.  The file is taken in one read and cut at the '>' that start a line: one split/join
    per record instead of one per line."""
    text = handle.read()
    if text.startswith('>'):
        text = '\n' + text
    for part in text.split('\n>')[1:]:       # (whatever precedes the first header is not a record)
        header, _, body = part.partition('\n')
        header = header.strip()
        yield (header.split()[0] if header else ''), ''.join(body.split())
