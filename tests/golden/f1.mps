NAME          F1
OBJSENSE
    MAX
ROWS
 N  obj
 L  c1
 L  c2
 L  c3
COLUMNS
    MARKER                 'MARKER'                 'INTORG'
    x1        obj                5.0   c1                 2.0
    x1        c2                 4.0   c3                 3.0
    x2        obj                4.0   c1                 3.0
    x2        c2                 1.0   c3                 4.0
    x3        obj                3.0   c1                 1.0
    x3        c2                 2.0   c3                 2.0
    x4        obj                7.0   c1                 4.0
    x4        c2                 3.0   c3                 1.0
    x5        obj                6.0   c1                 2.0
    x5        c2                 5.0   c3                 3.0
    MARKER                 'MARKER'                 'INTEND'
RHS
    RHS       c1                15.0   c2                23.0
    RHS       c3                17.0
ENDATA
