"""Minimal line-chart writer (SVG, no dependencies) for the bench drivers' --plot-flops: matplotlib is not part of the
ROCm image this repo targets; when it IS importable the drivers use it and write the reference's .png instead."""


def line_chart(path, title, xlabels, series, xlabel="M=N=K", ylabel="TFLOPS", width=1280, height=720):
    """series: list of (label, [y...], style) with style in {"solid", "dash", "bold"}; y may hold None gaps."""
    L, R, T, B = 70, 330, 50, 70
    pw, ph = width - L - R, height - T - B
    ys = [y for _, v, _ in series for y in v if y is not None]
    ymax = max(ys) * 1.05 if ys else 1.0
    n = max(len(xlabels), 2)
    X = lambda i: L + pw * i / (n - 1)
    Y = lambda y: T + ph * (1.0 - y / ymax)
    pal = ["#1f77b4", "#d62728", "#2ca02c", "#9467bd", "#8c564b", "#e377c2", "#7f7f7f", "#bcbd22", "#17becf", "#ff7f0e"]
    out = ['<svg xmlns="http://www.w3.org/2000/svg" width="%d" height="%d" font-family="sans-serif" font-size="12">' % (width, height),
           '<rect width="100%" height="100%" fill="white"/>',
           '<text x="%d" y="28" font-size="18" text-anchor="middle">%s</text>' % (L + pw // 2, title)]
    for g in range(6):
        yv = ymax * g / 5
        out.append('<line x1="%d" y1="%.1f" x2="%d" y2="%.1f" stroke="#ddd"/>' % (L, Y(yv), L + pw, Y(yv)))
        out.append('<text x="%d" y="%.1f" text-anchor="end">%.0f</text>' % (L - 6, Y(yv) + 4, yv))
    step = max(1, len(xlabels) // 25)
    for i, xl in enumerate(xlabels):
        if i % step == 0:
            out.append('<text x="%.1f" y="%d" text-anchor="end" transform="rotate(-45 %.1f %d)">%s</text>' % (X(i), T + ph + 16, X(i), T + ph + 16, xl))
    out.append('<rect x="%d" y="%d" width="%d" height="%d" fill="none" stroke="#333"/>' % (L, T, pw, ph))
    out.append('<text x="%d" y="%d" text-anchor="middle">%s</text>' % (L + pw // 2, height - 8, xlabel))
    out.append('<text x="16" y="%d" text-anchor="middle" transform="rotate(-90 16 %d)">%s</text>' % (T + ph // 2, T + ph // 2, ylabel))
    for si, (label, vals, style) in enumerate(series):
        col = pal[si % len(pal)]
        pts = " ".join("%.1f,%.1f" % (X(i), Y(y)) for i, y in enumerate(vals) if y is not None)
        sw = {"solid": 2, "dash": 1.5, "bold": 4}[style]
        dash = ' stroke-dasharray="6 4"' if style == "dash" else ""
        if len([y for y in vals if y is not None]) == 1:
            i, y = next((i, y) for i, y in enumerate(vals) if y is not None)
            out.append('<circle cx="%.1f" cy="%.1f" r="4" fill="%s"/>' % (X(i), Y(y), col))
        else:
            out.append('<polyline fill="none" stroke="%s" stroke-width="%s"%s points="%s"/>' % (col, sw, dash, pts))
        ly = T + 14 + si * 18
        out.append('<line x1="%d" y1="%d" x2="%d" y2="%d" stroke="%s" stroke-width="%s"%s/>' % (L + pw + 12, ly - 4, L + pw + 40, ly - 4, col, sw, dash))
        out.append('<text x="%d" y="%d">%s</text>' % (L + pw + 46, ly, label.replace("<", "&lt;").replace(">", "&gt;")))
    out.append("</svg>")
    with open(path, "w") as f:
        f.write("\n".join(out))
    return path
